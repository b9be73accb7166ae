cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "layernorm" 2>&1 | tail -5
python -m pytest tests/test_gpu_ddpg.py tests/test_gpu_agents.py -m gpu -q -x 2>&1 | tail -5
