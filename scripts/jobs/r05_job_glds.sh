#!/bin/bash
# A/B: LDS-DMA weight staging in layer 1 of the fused critic (libsurreal_amd_glds.so) vs the product library
mkdir -p gpurun_out
{
echo "== product"; for i in 1 2 3; do python scripts/bench_fused.py; done
echo "== glds"; for i in 1 2 3; do SMX_LIB_PATH=surreal_amd/libsurreal_amd_glds.so python scripts/bench_fused.py; done
echo "== glds tests"
SMX_LIB_PATH=surreal_amd/libsurreal_amd_glds.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused or mlp3" 2>&1 | tail -5
SMX_LIB_PATH=surreal_amd/libsurreal_amd_glds.so timeout 600 python -m pytest tests/test_gpu_learner.py -q -m gpu -k "cfg5 and not rnn" 2>&1 | tail -5
} > gpurun_out/r05_glds.log 2>&1
