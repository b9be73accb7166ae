#!/bin/bash
mkdir -p gpurun_out
SMX_ROWS_ONLY=1 SMX_DDPG_TBUF=1 SMX_LIB_PATH=$PWD/surreal_amd/libsurreal_amd_timing.so timeout 300 python scripts/bench_ddpg_rows.py > gpurun_out/ddpg_rows_phases.txt 2>&1
tail -32 gpurun_out/ddpg_rows_phases.txt
