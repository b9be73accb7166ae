#!/bin/bash
# round 6, final tree: the rollout kernel's rocprofv3 table, then the whole GPU tier, smoke(), the bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r06_rollout
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_rollout -o bench -- python $R/scripts/bench_rollout.py 1024 > $R/gpurun_out/r06_rollout.log 2>&1
f=$(find $R/gpurun_out/r06_rollout -name "*_kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r06_rollout_kernel_stats.csv
find $R/gpurun_out/r06_rollout -type f -delete
cd $R
head -3 gpurun_out/r06_rollout_kernel_stats.csv | cut -c1-160
bash scripts/jobs/r06_job_full.sh
