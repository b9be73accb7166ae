cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fold or lstm or wgrad" 2>&1 | tail -6
python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
python -m pytest tests/test_gpu_learner.py tests/test_gpu_dist.py -m gpu -q -x -k "rnn or two_rank_hip_learner_equals" 2>&1 | tail -6
