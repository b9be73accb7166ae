cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ddpg.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r05_ddpg_rows_phases.log
SMX_DDPG_TBUF=1 SMX_LIB_PATH=$PWD/surreal_amd/libsurreal_amd_ddpgt.so python scripts/bench_ddpg_rows.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_ddpg_rows_phases.log
