cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/prof_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_m.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lstm -o bench -- python scripts/bench_rnn_one.py 64 128 17 6 > gpurun_out/prof_lstm.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pix -o bench -- python scripts/bench_pixel_one.py > gpurun_out/prof_pix.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipe -o bench -- python scripts/bench_pipeline.py --graph --fused-step --actors 1024 --iters 5 > gpurun_out/prof_pipe.log 2>&1
ls gpurun_out/pmc_m gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/prof_b | head -30; tail -2 gpurun_out/prof_lstm.log gpurun_out/prof_pix.log gpurun_out/prof_pipe.log | cut -c1-300
