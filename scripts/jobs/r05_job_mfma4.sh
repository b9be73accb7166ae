cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "== 1024 x 128 LSTM, product"; python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
echo "== 4-row MFMA recurrences (round 2's kernels)"; SMX_LSTM_MFMA4=1 python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
} > gpurun_out/r05_mfma4.log 2>&1
