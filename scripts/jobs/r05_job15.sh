cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_epoch.py -m gpu -q -x -k "uneven_load" 2>&1 | grep -v amdgpu | tail -40
