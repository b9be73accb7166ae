# kernel table of the PPO 1024 x 128 LSTM learn on the current tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_rnn1024
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rnn1024 -o rnn -- python scripts/bench_rnn_one.py 1024 128 17 6 > gpurun_out/prof_rnn1024.log 2>&1
tail -1 gpurun_out/prof_rnn1024.log
f=$(find gpurun_out/prof_rnn1024 -name '*kernel_trace.csv' | head -1)
python scripts/trace_summary.py $f gpurun_out/r05_lstm_1024x128_kernel_stats_d.csv 'python scripts/bench_rnn_one.py 1024 128 17 6 (PPO 1024x128, LSTM policy; round-5 final tree)'
rm -rf gpurun_out/prof_rnn1024
