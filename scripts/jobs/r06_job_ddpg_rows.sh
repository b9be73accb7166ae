#!/bin/bash
# DDPG row-block schedule on 4-row workgroups: parity tests, timing against the level schedule, phase stamps
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ddpg.py -x -q -m gpu > gpurun_out/ddpg_rows_tests.txt 2>&1
tail -5 gpurun_out/ddpg_rows_tests.txt
timeout 300 python scripts/bench_ddpg_rows.py > gpurun_out/ddpg_rows_bench.txt 2>&1
SMX_DDPG_TBUF=1 SMX_LIB_PATH=$PWD/surreal_amd/libsurreal_amd_timing.so timeout 300 python scripts/bench_ddpg_rows.py > gpurun_out/ddpg_rows_phases.txt 2>&1
cat gpurun_out/ddpg_rows_bench.txt; tail -30 gpurun_out/ddpg_rows_phases.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rollout" 2>&1 | tail -3
