cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "wgrad or splitk or lstm" 2>&1 | tail -4
python -m pytest tests/test_gpu_learner.py tests/test_gpu_sequences.py -q -m gpu -x -k "rnn or lstm or sequence" 2>&1 | tail -4
echo "== 64 x 128 LSTM"; python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
echo "== the same, MLP weight gradients in front of the recurrence"; SMX_BENCH_LEARNER_OPTS=overlap_stem_wgrads=0 python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
echo "== 128 x 128 LSTM"; python scripts/bench_rnn_one.py 128 128 17 6 2>&1 | tail -1
echo "== the same, in front"; SMX_BENCH_LEARNER_OPTS=overlap_stem_wgrads=0 python scripts/bench_rnn_one.py 128 128 17 6 2>&1 | tail -1
} > gpurun_out/r05_wpair.log 2>&1
