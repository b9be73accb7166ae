cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipe -o bench -- python scripts/bench_pipeline.py --fused-step --iters 3 --warmup 1 > gpurun_out/prof_pipe.log 2>&1
tail -1 gpurun_out/prof_pipe.log | cut -c1-200
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_pipe/bench_kernel_stats.csv')))
for r in rows[:10]:
    print('%-70s calls %6s avg %9.2f us  %5s %%' % (r['Name'].replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
