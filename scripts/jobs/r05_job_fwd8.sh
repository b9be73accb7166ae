cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "== tests with the 16-wave forward"; SMX_LSTM_FWD8=1 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -3
SMX_LSTM_FWD8=1 python -m pytest tests/test_gpu_learner.py -q -m gpu -x -k "rnn" 2>&1 | tail -3
echo "== 64 x 128 LSTM, product"; python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
echo "== 16-wave forward"; SMX_LSTM_FWD8=1 python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
echo "== 256 x 128, product"; python scripts/bench_rnn_one.py 256 128 17 6 2>&1 | tail -1
echo "== 16-wave forward"; SMX_LSTM_FWD8=1 python scripts/bench_rnn_one.py 256 128 17 6 2>&1 | tail -1
} > gpurun_out/r05_fwd8.log 2>&1
