cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "lstm" 2>&1 | tail -6
python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
SMX_LSTM_NOFOLD=1 python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
SMX_LSTM_NOFOLD=1 python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
python -m pytest tests/test_gpu_learner.py tests/test_gpu_agents.py tests/test_gpu_sequences.py -m gpu -q -x 2>&1 | tail -6
