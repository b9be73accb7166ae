cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -4 > gpurun_out/r05_t1.log
python -m pytest tests/test_gpu_agents.py -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/r05_t1.log
