"""Turn the rocprofv3 CSV outputs under gpurun_out/ into the committed summaries under profiles/:
    <tag>_bench_kernel_stats.csv   per-kernel and per-grid-size durations (--kernel-trace --stats)
    <tag>_pmc_hbm_traffic.json     FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes)
    <tag>_pmc_mfma.json            MFMA-busy / issued MFMA flops per launch (its own --pmc pass)
usage: python scripts/profiles_summary.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir> [<pmc_mfma_dir>]]"""
import collections
import csv
import json
import sys


def short(name):
    if 'at::native' in name:
        return 'torch:' + name.split('at::native::')[1][:60].replace(',', ';')
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(',', ';')


N_XCD, N_SIMD = 8, 1024          # MI355X: 8 XCDs, 256 CUs x 4 SIMDs


def mfma_summary(tag, mfma_dir):
    """MfmaUtil as counter_defs.yaml defines it: sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE *
    SIMD_NUM).  The CSV holds each counter summed over its instances, so GRBM_GUI_ACTIVE (one per
    XCD) is divided by the 8 XCDs to get the kernel's active cycles."""
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(mfma_dir + '/bench_counter_collection.csv')):
        n = short(r['Kernel_Name'])
        if any(t in n for t in ('fused', 'rows16', 'gemm', 'lstm', 'epoch', 'adam', 'bwd16', 'wgrad')):
            a[(n, int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {'note': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES '
                   'SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE (own pass), python bench.py --steps 3 --warmup 1 '
                   '--no-graph --no-cpu-baseline.  Means per launch.  mfma_util_pct = MFMA_BUSY / '
                   '(GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) * 100 (the MfmaUtil formula of counter_defs.yaml); '
                   'mfma_flops = MOPS_F32 * 512; active clock = GRBM_GUI_ACTIVE / 8 / kernel time is what the '
                   'chip ran at under the profiler (DVFS), so util and the bench roofline.frac (priced at the '
                   '2.4 GHz peak) differ by that clock ratio.',
           'kernels': {}}
    for (n, g), c in sorted(a.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        gui = m.get('GRBM_GUI_ACTIVE', 0.0) / N_XCD
        e = {'launches': len(next(iter(c.values()))), 'counters_mean': m}
        if gui:
            e['active_cycles_per_xcd'] = gui
            e['mfma_util_pct'] = 100.0 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * N_SIMD)
        e['mfma_flops'] = m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) * 512
        res['kernels']['%s@grid%d' % (n, g)] = e
    json.dump(res, open('profiles/%s_pmc_mfma.json' % tag, 'w'), indent=1)
    for k, e in res['kernels'].items():
        print('%-46s util %5.1f %%  flops %.3e' % (k, e.get('mfma_util_pct', float('nan')), e['mfma_flops']))


def main(tag, stats_dir, fetch_dir=None, write_dir=None, mfma_dir=None):
    if mfma_dir:
        mfma_summary(tag, mfma_dir)
    rows = list(csv.DictReader(open(stats_dir + '/bench_kernel_stats.csv')))
    out = 'profiles/%s_bench_kernel_stats.csv' % tag
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py '
                '--steps 10 --warmup 3 --no-cpu-baseline --no-secondary   (one MI355X)\n')
        f.write('kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n')
        for r in rows:
            f.write('%s,%s,%s,%.1f,%s,%s,%s\n' % (short(r['Name']), r['Calls'], r['TotalDurationNs'],
                                                  float(r['AverageNs']), r['Percentage'], r['MinNs'], r['MaxNs']))
        tr = list(csv.DictReader(open(stats_dir + '/bench_kernel_trace.csv')))
        agg = collections.defaultdict(list)
        for r in tr:
            n = short(r['Kernel_Name'])
            if 'gemm' in n or 'fused' in n or 'rows16' in n or 'epoch' in n or 'adam' in n:
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
                  if 'fused' in k[0] or 'rows16' in k[0] or 'gemm' in k[0] or 'adam' in k[0] or 'epoch' in k[0]}
    res = {'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (--kernel-trace only), '
                   'python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline.  Counter means per launch in KiB '
                   'as reported.  On gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced '
                   'streaming reads (MI355X_MICROARCH.md, HBM section): hbm_read_bytes = 2 * FETCH_SIZE * 1024.  '
                   'WRITE_SIZE is taken as reported (it equals the 4 B x rows the kernel writes).',
           'counters': pmc, 'fused_kernel': {}}
    for k in pmc['FETCH_SIZE']:
        if 'fused' not in k and 'rows16' not in k:
            continue
        # 128 rows per workgroup: 256 threads (32-row wavefronts) or 512 (16-row wavefronts, smx_mlp3_rows16.hip)
        nrows = int(k.split('grid')[1]) // (512 if 'rows16' in k else 256) * 128
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
