cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "many_rows or splitk or lstm" 2>&1 | tail -12
python - <<'PY'
import sys; sys.path.insert(0, '.')
exec(open('scripts/bench_stem_mlp.py').read().split("fl = 2.0 * rows")[0])
packT = f(K.mlp3_dgrad_rows_ws_floats(net)); dx = f(rows, D)
t = timed(lambda: K.mlp3_backward(net, x, h1, h2, dz3, dz2, dz1, grads, None, ws=ws, packT=packT, dx=dx))
print('backward fused dgrad (+dx) + wgrad: %8.1f us' % t)
t = timed(lambda: (K.mlp3_backward(net, x, h1, h2, dz3, dz2, dz1, grads, None, ws=ws), K.linear(dz1, 1, net.views['W1'], 0, None, dx, rows, D, H1)))
print('backward layered dgrad + dx + wgrad: %8.1f us' % t)
PY
python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
python -m pytest tests/test_gpu_learner.py -m gpu -q -x -k "rnn" 2>&1 | tail -6
