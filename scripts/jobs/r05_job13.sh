cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or splitk or lstm or fused_data" 2>&1 | tail -4
python scripts/bench_rnn_one.py 1024 128 376 17 2>&1 | tail -1
SMX_WGRAD_TILED=1 python scripts/bench_rnn_one.py 1024 128 376 17 2>&1 | tail -1
python -m pytest tests/test_gpu_learner.py -m gpu -q -x -k "rnn" 2>&1 | tail -3
