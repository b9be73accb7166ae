cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -3
python -m pytest tests/test_gpu_learner.py tests/test_gpu_sequences.py -q -m gpu -x -k "rnn or lstm or sequence" 2>&1 | tail -3
echo "== 64 x 128 LSTM"; python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
echo "== 256 x 128 LSTM"; python scripts/bench_rnn_one.py 256 128 17 6 2>&1 | tail -1
echo "== 256 x 32 pixel"; python scripts/bench_pixel_one.py 2>&1 | tail -1
} > gpurun_out/r05_bwdk.log 2>&1
