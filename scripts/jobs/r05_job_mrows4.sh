cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -3
echo "== 1024 x 128 LSTM"; python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
echo "== cfg5 LSTM"; python scripts/bench_rnn_one.py 1024 128 376 17 2>&1 | tail -1
} > gpurun_out/r05_mrows4.log 2>&1
