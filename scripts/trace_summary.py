"""rocprofv3 --kernel-trace CSV -> the per-(kernel, grid) summary committed under profiles/
usage: python scripts/trace_summary.py <bench_kernel_trace.csv> <out.csv> "<what was run>" """
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    if 'at::native' in n:
        return 'torch:' + n.split('at::native::')[1][:60].replace(',', ';')
    return n.split('(')[0].replace(',', ';')


def main(trace, out, what):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(trace)):
        k = (short(r['Kernel_Name']), int(r['Grid_Size_X']))
        agg[k][0] += 1
        agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(v[1] for v in agg.values())
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --output-format csv -- %s\n' % what)
        f.write('kernel,grid_x,calls,avg_us,total_ms,pct\n')
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if v[1] / tot < 0.001:
                continue
            f.write('%s,%d,%d,%.1f,%.2f,%.2f\n' % (k[0], k[1], v[0], v[1] / v[0] / 1e3, v[1] / 1e6, 100.0 * v[1] / tot))
    print('wrote', out, '(%.2f ms of kernels)' % (tot / 1e6))


if __name__ == '__main__':
    main(*sys.argv[1:4])
