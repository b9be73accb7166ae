"""Turn the rocprofv3 CSV outputs under gpurun_out/ into the committed summaries under profiles/:
    <tag>_bench_kernel_stats.csv   per-kernel and per-grid-size durations (--kernel-trace --stats)
    <tag>_pmc_hbm_traffic.json     FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes)
usage: python scripts/profiles_summary.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]"""
import collections
import csv
import json
import sys


def short(name):
    if 'at::native' in name:
        return 'torch:' + name.split('at::native::')[1][:60].replace(',', ';')
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(',', ';')


def main(tag, stats_dir, fetch_dir=None, write_dir=None):
    rows = list(csv.DictReader(open(stats_dir + '/bench_kernel_stats.csv')))
    out = 'profiles/%s_bench_kernel_stats.csv' % tag
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py '
                '--steps 10 --warmup 3 --no-cpu-baseline   (one MI355X)\n')
        f.write('kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n')
        for r in rows:
            f.write('%s,%s,%s,%.1f,%s,%s,%s\n' % (short(r['Name']), r['Calls'], r['TotalDurationNs'],
                                                  float(r['AverageNs']), r['Percentage'], r['MinNs'], r['MaxNs']))
        tr = list(csv.DictReader(open(stats_dir + '/bench_kernel_trace.csv')))
        agg = collections.defaultdict(list)
        for r in tr:
            n = short(r['Kernel_Name'])
            if 'gemm' in n or 'fused' in n:
                agg[(n, int(r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        f.write('\n# per grid size (threads): kernel,grid_x,calls,avg_ns,min_ns\n')
        for (n, g), v in sorted(agg.items()):
            f.write('%s,%d,%d,%.0f,%d\n' % (n, g, len(v), sum(v) / len(v), min(v)))
            print('%-40s grid %7d  calls %4d  avg %8.1f us  min %8.1f us' % (n, g, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
    if not fetch_dir:
        return
    pmc = {}
    for c, d in (('FETCH_SIZE', fetch_dir), ('WRITE_SIZE', write_dir)):
        a = collections.defaultdict(list)
        for r in csv.DictReader(open(d + '/bench_counter_collection.csv')):
            if r['Counter_Name'] == c:
                a[(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
        pmc[c] = {'%s@grid%d' % k: {'launches': len(v), 'mean_kib': sum(v) / len(v)} for k, v in a.items()
                  if 'fused' in k[0] or 'gemm' in k[0] or 'adam' in k[0]}
    res = {'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (--kernel-trace only), '
                   'python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline.  Counter means per launch in KiB '
                   'as reported.  On gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced '
                   'streaming reads (MI355X_MICROARCH.md, HBM section): hbm_read_bytes = 2 * FETCH_SIZE * 1024.  '
                   'WRITE_SIZE is taken as reported (it equals the 4 B x rows the kernel writes).',
           'counters': pmc, 'fused_kernel': {}}
    for k in pmc['FETCH_SIZE']:
        if 'fused' not in k:
            continue
        nrows = int(k.split('grid')[1]) // 256 * 128
        fetch = pmc['FETCH_SIZE'][k]['mean_kib'] * 1024 * 2
        wr = pmc['WRITE_SIZE'].get(k, {'mean_kib': 0})['mean_kib'] * 1024
        alg = nrows * 377 * 4.0
        res['fused_kernel'][k] = {'rows_upper_bound': nrows, 'hbm_read_bytes_corrected': fetch,
                                  'hbm_write_bytes': wr, 'algorithmic_bytes': alg,
                                  'traffic_over_algorithmic': (fetch + wr) / alg}
    json.dump(res, open('profiles/%s_pmc_hbm_traffic.json' % tag, 'w'), indent=1)
    print(json.dumps(res['fused_kernel'], indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:])
