cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "many_rows or splitk or lstm" 2>&1 | tail -8
echo "== stem MLP, 126976 x 100 -> 300 -> 200 -> 6"; python scripts/bench_stem_mlp.py 2>&1 | grep -v amdgpu.ids
python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
python scripts/bench_rnn_one.py 64 128 17 6 2>&1 | tail -1
python -m pytest tests/test_gpu_learner.py -m gpu -q -x -k "rnn or timeout" 2>&1 | tail -8
