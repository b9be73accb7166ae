cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import bench_rnn
bench_rnn.run(64, 128, 17, 6, steps=3)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rnn -o rnn -- python /tmp/one.py > gpurun_out/prof_rnn.log 2>&1
tail -2 gpurun_out/prof_rnn.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_rnn/rnn_kernel_stats.csv')))
for r in rows[:14]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]
    print('%-70s calls %5s avg %9.1f us pct %s' % (n, r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
