cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pix -o bench -- python scripts/bench_pixel_one.py > gpurun_out/prof_pix.log 2>&1
tail -2 gpurun_out/prof_pix.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_pix/bench_kernel_stats.csv')))
for r in rows[:14]:
    print('%-70s calls %5s avg %9.1f us  %5s %%' % (r['Name'].replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
