# GPU call 3 of round 5: new kernels (fused many-row forward, wgrad from row-major operands), co-tenant diagnostics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "many_rows or splitk" 2>&1 | tail -15
echo "== stem MLP, 126976 x 100 -> 300 -> 200 -> 6"; python scripts/bench_stem_mlp.py 2>&1 | grep -v amdgpu.ids
echo "== the same, weight gradients on the tiled GEMM"; SMX_WGRAD_TILED=1 python scripts/bench_stem_mlp.py 2>&1 | grep -v amdgpu.ids
echo "== 126976 x 100 -> 17 (cfg5 LSTM)"; python scripts/bench_stem_mlp.py 126976 100 17 2>&1 | grep -v amdgpu.ids
echo "== 7168 x 288 -> 8 (cfg3 pixel)"; python scripts/bench_stem_mlp.py 7168 288 8 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import sys, time; sys.path.insert(0, '.')
import torch
from surreal_amd.kernels import HipKernels
K = HipKernels()
for blocks, us in ((16, 100000), (224, 450000)):
    torch.cuda.synchronize(); t0 = time.time()
    K.device_occupy(blocks, us); torch.cuda.synchronize()
    print('occupy(%d blocks, %d us) took %.1f ms' % (blocks, us, (time.time() - t0) * 1e3))
PY
python -m pytest tests/test_gpu_learner.py -m gpu -q -x -k "timeout or co_resident or shared_device" 2>&1 | tail -15
python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -3
