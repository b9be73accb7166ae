cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "wgrad_pair" 2>&1 | grep -v amdgpu | tail -40 > gpurun_out/r05_wpair2.log 2>&1
