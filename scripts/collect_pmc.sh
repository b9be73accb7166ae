cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_f.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_w.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_m.log 2>&1
ls gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m | head -20
