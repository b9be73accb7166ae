#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ddpg.py tests/test_gpu_dist.py -x -q -m gpu -k ddpg > gpurun_out/ddpg_rows_tests.txt 2>&1
tail -4 gpurun_out/ddpg_rows_tests.txt
timeout 300 python scripts/bench_ddpg.py 2>&1 | tail -5
timeout 300 python scripts/prof_ddpg_host.py 2>&1 | grep -v "^$" | sed -n 2,22p | cut -c1-150
