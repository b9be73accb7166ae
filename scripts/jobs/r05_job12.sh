cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fused_data_gradients" 2>&1 | tail -4
for v in "" nohoist simple; do
  if [ -z "$v" ]; then lib=surreal_amd/libsurreal_amd.so; else lib=surreal_amd/libsurreal_amd_$v.so; fi
  echo "== ${v:-product (interleaved + hoist)}"; SMX_LIB_PATH=$PWD/$lib python scripts/bench_stem_mlp.py 2>&1 | grep fused
done
python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
python scripts/bench_rnn_one.py 1024 128 376 17 2>&1 | tail -1
