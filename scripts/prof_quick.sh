# quick per-kernel time table of one bench.py run (rocprofv3 --kernel-trace --stats); args are passed to bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/prof_q
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" > gpurun_out/prof_q.log 2>&1
tail -1 gpurun_out/prof_q.log | cut -c1-330
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_q/**/bench_kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    agg[(n, int(r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print('%-62s %8s %6s %9s %6s' % ('kernel', 'grid', 'calls', 'avg_us', 'pct'))
for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-62s %8d %6d %9.2f %6.2f' % (n, g, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot))
PY
