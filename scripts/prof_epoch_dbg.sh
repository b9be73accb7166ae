# device time of the fused epoch kernels under the SMX_EPOCH_DBG timing experiments (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for dbg in "$@"; do
  rm -rf gpurun_out/prof_d
  SMX_EPOCH_DBG_ONLY=$dbg timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_d -o e -- python scripts/bench_epoch.py only > gpurun_out/prof_d.log 2>&1
  python - "$dbg" <<'PY'
import csv, glob, collections, sys
f = glob.glob('gpurun_out/prof_d/**/e_kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-40:]
    agg[(n, int(r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('dbg', sys.argv[1], ' '.join('%s[%d] %.2f us (min %.2f)' % (n[:22], g, sum(v[5:]) / len(v[5:]) / 1e3, min(v) / 1e3)
                                 for (n, g), v in sorted(agg.items()) if 'epoch' in n or 'gemm' in n))
PY
done
