cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
for b in 512 768; do
echo "== $b x 128, vector"; python scripts/bench_rnn_one.py $b 128 17 6 2>&1 | tail -1
echo "== $b x 128, matrix-pipe"; SMX_LSTM_MROWS_MIN=256 python scripts/bench_rnn_one.py $b 128 17 6 2>&1 | tail -1
done
echo "== 256 x 128, matrix-pipe"; SMX_LSTM_MROWS_MIN=256 python scripts/bench_rnn_one.py 256 128 17 6 2>&1 | tail -1
} > gpurun_out/r05_mrows3.log 2>&1
