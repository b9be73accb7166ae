#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ddpg.py -x -q -m gpu > gpurun_out/ddpg_rows_tests.txt 2>&1
tail -15 gpurun_out/ddpg_rows_tests.txt
