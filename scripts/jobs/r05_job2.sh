# GPU call 2 of round 5: full GPU tier, 1024x128 LSTM kernel table, fused-critic ablations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r05_gputest2.log 2>&1; tail -4 gpurun_out/r05_gputest2.log
for v in "" prio1 exp1 exp2 exp4 exp8 exp7 exp15; do
  if [ -z "$v" ]; then lib=surreal_amd/libsurreal_amd.so; else lib=surreal_amd/libsurreal_amd_$v.so; fi
  echo "== variant ${v:-product}" ; SMX_LIB_PATH=$PWD/$lib timeout 120 python scripts/bench_fused.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_fused_ablation.log 2>&1
cat gpurun_out/r05_fused_ablation.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rnn1024 -o rnn -- python scripts/bench_rnn_one.py 1024 128 17 6 > gpurun_out/prof_rnn1024.log 2>&1
tail -3 gpurun_out/prof_rnn1024.log
f=$(find gpurun_out/prof_rnn1024 -name '*kernel_trace.csv' | head -1)
python scripts/trace_summary.py $f gpurun_out/r05_lstm_1024x128_kernel_stats.csv 'python scripts/bench_rnn_one.py 1024 128 17 6 (PPO 1024x128, LSTM policy; round-5 tree before the fused stem MLP)'
head -45 gpurun_out/r05_lstm_1024x128_kernel_stats.csv
rm -rf gpurun_out/prof_rnn1024
