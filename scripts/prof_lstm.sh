cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lstm -o bench -- python scripts/bench_rnn_one.py 64 128 17 6 > gpurun_out/prof_lstm.log 2>&1
tail -5 gpurun_out/prof_lstm.log
head -30 gpurun_out/prof_lstm/bench_kernel_stats.csv | cut -c1-160
