cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r05_gputest3.log 2>&1; tail -4 gpurun_out/r05_gputest3.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rnn1024 -o rnn -- python scripts/bench_rnn_one.py 1024 128 17 6 > gpurun_out/prof_rnn1024.log 2>&1
tail -1 gpurun_out/prof_rnn1024.log
f=$(find gpurun_out/prof_rnn1024 -name '*kernel_trace.csv' | head -1)
python scripts/trace_summary.py $f gpurun_out/r05_lstm_1024x128_kernel_stats_c.csv 'python scripts/bench_rnn_one.py 1024 128 17 6 (PPO 1024x128, LSTM policy; round-5 tree: fused stem forward + data gradients, register-resident wgrad, 2 / 4-row LSTM workgroups, folded input projection)'
head -36 gpurun_out/r05_lstm_1024x128_kernel_stats_c.csv
rm -rf gpurun_out/prof_rnn1024
(time python bench.py > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err); tail -c 4000 gpurun_out/r05_bench2.json
