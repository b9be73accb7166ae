#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ddpg.py tests/test_gpu_dist.py -x -q -m gpu -k ddpg > gpurun_out/ddpg_rows_tests.txt 2>&1
tail -15 gpurun_out/ddpg_rows_tests.txt
timeout 300 python scripts/bench_ddpg.py > gpurun_out/ddpg_bench.txt 2>&1
tail -4 gpurun_out/ddpg_bench.txt
