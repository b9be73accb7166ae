# round 5, final tree: the GPU tier, the bench line, the full record, rocprofv3 kernel tables and PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r05_gputest_final.log 2>&1; tail -3 gpurun_out/r05_gputest_final.log
(time python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err); tail -c 600 gpurun_out/r05_bench_final.json; cp gpurun_out/bench_full.json gpurun_out/r05_bench_full_default.json
python bench.py --secondary all --budget-s 420 --full-out gpurun_out/r05_bench_full_all.json > gpurun_out/r05_bench_all_line.json 2> gpurun_out/r05_bench_all.err; tail -c 300 gpurun_out/r05_bench_all_line.json
rm -rf gpurun_out/prof_b gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m gpurun_out/prof_lstm gpurun_out/pmc_stem gpurun_out/prof_rnn1024
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/prof_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_m.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_stem -o bench -- python scripts/bench_rnn_one.py 1024 128 17 6 > gpurun_out/pmc_stem.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lstm -o bench -- python scripts/bench_rnn_one.py 64 128 17 6 > gpurun_out/prof_lstm.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rnn1024 -o rnn -- python scripts/bench_rnn_one.py 1024 128 17 6 > gpurun_out/prof_rnn1024.log 2>&1
f=$(find gpurun_out/prof_rnn1024 -name '*kernel_trace.csv' | head -1)
python scripts/trace_summary.py $f gpurun_out/r05_lstm_1024x128_kernel_stats_final.csv 'python scripts/bench_rnn_one.py 1024 128 17 6 (PPO 1024x128, LSTM policy; round-5 final tree)'
f=$(find gpurun_out/prof_lstm -name '*kernel_trace.csv' | head -1)
python scripts/trace_summary.py $f gpurun_out/r05_lstm_cfg1_kernel_stats_final.csv 'python scripts/bench_rnn_one.py 64 128 17 6 (configs[1] shapes, LSTM-stem policy, the reference default; round-5 final tree)'
rm -rf gpurun_out/prof_rnn1024
# flatten rocprofv3's per-host sub-directory so that scripts/profiles_summary.py finds the files
for d in prof_b pmc_f pmc_w pmc_m pmc_stem prof_lstm; do find gpurun_out/$d -mindepth 2 -name '*.csv' -exec mv {} gpurun_out/$d/ \; ; done
find gpurun_out/prof_b gpurun_out/pmc_stem -maxdepth 1 -name '*.csv' | head; du -sh gpurun_out/pmc_* gpurun_out/prof_*
