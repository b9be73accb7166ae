#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*: run on the GPU box from the repository root (gpurun), then
#   python scripts/profiles_summary.py <tag> gpurun_out/<tag>_stats gpurun_out/<tag>_fetch gpurun_out/<tag>_write gpurun_out/<tag>_mfma
# turns them into the committed summaries.  Counters are collected in their own passes (--kernel-trace only beside them).
# usage: bash scripts/profile_round.sh <tag>
TAG=${1:-r06}
R=$(pwd)
export TMPDIR=/tmp
norm() {   # rocprofv3 nests its outputs under <dir>/<host>/...: flat copies under the names the summary script reads
    for k in kernel_stats kernel_trace counter_collection; do
        f=$(find "$1" -name "*_${k}.csv" | head -1)
        [ -n "$f" ] && cp "$f" "$1/bench_${k}.csv"
    done
}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o bench -- python $R/bench.py --no-secondary --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
norm $R/gpurun_out/${TAG}_stats
PMC="python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -o bench -- $PMC > $R/gpurun_out/${TAG}_fetch.log 2>&1
norm $R/gpurun_out/${TAG}_fetch
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -o bench -- $PMC > $R/gpurun_out/${TAG}_write.log 2>&1
norm $R/gpurun_out/${TAG}_write
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_mfma -o bench -- $PMC > $R/gpurun_out/${TAG}_mfma.log 2>&1
norm $R/gpurun_out/${TAG}_mfma
# the rollout kernel alone
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_rollout -o bench -- python $R/scripts/bench_rollout.py 1024 > $R/gpurun_out/${TAG}_rollout.log 2>&1
norm $R/gpurun_out/${TAG}_rollout
cd $R
# keep what travels back small: the flat copies only
find gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma gpurun_out/${TAG}_rollout -type f ! -name "bench_*.csv" -delete 2>/dev/null
ls -la gpurun_out/${TAG}_*/ | head -40
