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
python - <<'PY'
import csv, collections
tr=list(csv.DictReader(open('gpurun_out/prof_pix/bench_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in tr:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    agg[(n,int(r['Grid_Size_X']))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=sum(sum(v) for v in agg.values())
for (n,g),v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:14]:
    print('%-42s grid %9d calls %4d avg %8.1f us  total/learn %6.1f ms (%.1f%%)'%(n,g,len(v),sum(v)/len(v)/1e3,sum(v)/1e6/4, 100*sum(v)/tot))
PY
