#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ddpg.py -x -q -m gpu > gpurun_out/ddpg_rows_tests.txt 2>&1
tail -5 gpurun_out/ddpg_rows_tests.txt
timeout 300 python scripts/bench_ddpg.py > gpurun_out/ddpg_bench.txt 2>&1
tail -4 gpurun_out/ddpg_bench.txt
bash scripts/jobs/r06_job_ddpg_rows_prof.sh 2>&1 | cut -c1-160 | head -8
bash scripts/jobs/r06_job_ddpg_rows4.sh | tail -1 | cut -c1-420
