#!/usr/bin/env python
"""profiles/<tag>_pmc_hbm_small.json from three rocprofv3 passes over scripts/bench_small_kernels.py:
    python scripts/profiles_hbm_small.py <tag> <stats_dir> <fetch_dir> <write_dir> <bench_stdout_log>
Per kernel: average duration (kernel trace), FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes; FETCH_SIZE x 2:
the gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md HBM section), GB/s of the measured traffic and of
the algorithmic bytes against the 8 TB/s peak."""
import collections
import csv
import glob
import json
import sys

PEAK = 8000.0


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]


def find(d, pat):
    return glob.glob(d + '/**/' + pat, recursive=True)[0]


def main(tag, stats_dir, fetch_dir, write_dir, log):
    alg = None
    for ln in open(log):
        if ln.startswith('{') and 'algorithmic_bytes_per_launch' in ln:
            alg = json.loads(ln)['algorithmic_bytes_per_launch']
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(find(stats_dir, '*kernel_trace.csv'))):
        dur[(short(r['Kernel_Name']), int(r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    ctr = {}
    for which, d in (('FETCH_SIZE', fetch_dir), ('WRITE_SIZE', write_dir)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(find(d, '*counter_collection.csv'))):
            if r['Counter_Name'] == which:
                acc[(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
        ctr[which] = {k: sum(v) / len(v) for k, v in acc.items()}
    out = {'note': 'scripts/bench_small_kernels.py under rocprofv3: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE '
                   '(three separate passes).  hbm_read_bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 reports half of the bytes of '
                   'wide coalesced reads: MI355X_MICROARCH.md, HBM section; narrow / strided reads are uncalibrated there and may be '
                   'over-corrected by the factor 2), hbm_write_bytes = WRITE_SIZE KiB x 1024.  GBps_algorithmic prices the '
                   'bytes that must move once (scripts/bench_small_kernels.py) against the same duration; peak 8 TB/s.',
           'kernels': {}}
    for (name, grid), v in sorted(dur.items()):
        keys = [k for k in (alg or {}) if k.split('@')[0] == name]
        if not keys:
            continue
        us = sum(v) / len(v) / 1e3
        f = ctr['FETCH_SIZE'].get((name, grid))
        w = ctr['WRITE_SIZE'].get((name, grid))
        e = {'launches': len(v), 'avg_us': us}
        if f is not None and w is not None:
            rb, wb = 2.0 * f * 1024, w * 1024
            e.update(hbm_read_bytes=rb, hbm_write_bytes=wb, GBps_measured=(rb + wb) / us / 1e3,
                     frac_of_hbm_peak_measured=(rb + wb) / us / 1e3 / PEAK)
        # (zupdate runs at two shapes: match by duration order -- the larger algorithmic figure to the longer launch)
        key = keys[0] if len(keys) == 1 else None
        out['kernels']['%s@grid%d' % (name, grid)] = e
        e['_keys'] = keys
    # assign algorithmic bytes: unique names directly; a kernel run at several shapes ("name@rows x cols") by ascending
    # grid size <-> ascending column count (zupdate's grid grows with the columns)
    byname = collections.defaultdict(list)
    for k, e in out['kernels'].items():
        byname[k.split('@')[0]].append((int(k.split('@grid')[1]), k))
    for name, lst in byname.items():
        keys = sorted(((int(k.split('x')[-1]) if '@' in k else 0, alg[k], k) for k in alg if k.split('@')[0] == name))
        keys = [(b, k) for _, b, k in keys]
        for (g, k), (b, ak) in zip(sorted(lst), keys):
            us = out['kernels'][k]['avg_us']
            e = out['kernels'][k]
            e.pop('_keys', None)
            e.update(algorithmic_bytes=b, shape=ak, GBps_algorithmic=b / us / 1e3, frac_of_hbm_peak_algorithmic=b / us / 1e3 / PEAK)
            if 'hbm_read_bytes' in e:
                e['traffic_over_algorithmic'] = (e['hbm_read_bytes'] + e['hbm_write_bytes']) / b
    for e in out['kernels'].values():
        e.pop('_keys', None)
    json.dump(out, open('profiles/%s_pmc_hbm_small.json' % tag, 'w'), indent=1)
    for k, e in out['kernels'].items():
        print('%-44s %8.2f us  alg %7.1f GB/s (%.3f)  measured %s' % (
            k, e['avg_us'], e.get('GBps_algorithmic', float('nan')), e.get('frac_of_hbm_peak_algorithmic', float('nan')),
            '%.1f GB/s x%.2f' % (e['GBps_measured'], e.get('traffic_over_algorithmic', float('nan'))) if 'GBps_measured' in e else '-'))


if __name__ == '__main__':
    main(*sys.argv[1:6])
