# kernel table of the cfg-5 LSTM learn (1024 x 128 x 376, A = 17)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cfg5
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -o rnn -- python scripts/bench_rnn_one.py 1024 128 376 17 > gpurun_out/prof_cfg5.log 2>&1
tail -1 gpurun_out/prof_cfg5.log
f=$(find gpurun_out/prof_cfg5 -name '*kernel_trace.csv' | head -1)
python scripts/trace_summary.py $f gpurun_out/r05_lstm_cfg5_kernel_stats.csv 'python scripts/bench_rnn_one.py 1024 128 376 17 (cfg-5 shapes with the LSTM policy)'
rm -rf gpurun_out/prof_cfg5
