#!/bin/bash
# kernel table of one DDPG iteration on the row-block schedule
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SMX_ROWS_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ddpg_rows_prof -o t -- python $R/scripts/bench_ddpg_rows.py > $R/gpurun_out/ddpg_rows_prof.log 2>&1
cd $R
f=$(find gpurun_out/ddpg_rows_prof -name "*_kernel_stats.csv" | head -1); cp "$f" gpurun_out/ddpg_rows_kernel_stats.csv; find gpurun_out/ddpg_rows_prof -type f -delete
head -25 gpurun_out/ddpg_rows_kernel_stats.csv
