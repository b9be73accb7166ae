#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/cfg1mlp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg1mlp -o t -- python $R/scripts/prof_cfg1_mlp.py > $R/gpurun_out/cfg1mlp.log 2>&1
cd $R
tail -2 gpurun_out/cfg1mlp.log | cut -c1-300
f=$(find gpurun_out/cfg1mlp -name "*_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print('%-70s calls %5s avg %9.1f ns  %5s %%' % (r['Name'].replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs']), r['Percentage']))
PY
find gpurun_out/cfg1mlp -type f -delete
