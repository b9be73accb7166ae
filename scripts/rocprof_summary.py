"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as text:
per-kernel calls / total / average duration, and per-grid-size averages for the GEMM kernel."""
import sqlite3
import sys


def short_name(name):
    if 'at::native' in name:
        return 'torch:' + name.split('at::native::')[1][:70]
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0]


def main(db, out):
    c = sqlite3.connect(db)
    lines = ['# rocprofv3 --kernel-trace --stats summary of %s' % db,
             '# columns: calls  total_us  avg_us  pct  kernel', '']
    for name, calls, total, avg, pct in c.execute(
            'select name, total_calls, total_duration, average, percentage from top_kernels'):
        lines.append('%6d %12.1f %10.3f %6.2f  %s' % (calls, total, avg, pct, short_name(name)))
    lines += ['', '# GEMM kernels by grid size (threads): kernel grid_x  calls  avg_us  min_us']
    for r in c.execute("select name, grid_x, count(*), avg(end-start)/1000.0, min(end-start)/1000.0 "
                       "from kernels where name like '%gemm%' group by name, grid_x order by name, grid_x"):
        lines.append('%s %8d %6d %9.3f %9.3f' % ((short_name(r[0]),) + tuple(r[1:])))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:16]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
