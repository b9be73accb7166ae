"""time the fused critic pass (131 072 step rows x 376 -> 300 -> 200 -> 1, z-filter on) on the GPU box;
SMX_FUSED32=1 python scripts/bench_fused.py keeps the 32-row kernel for comparison"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from surreal_amd.kernels import HipKernels
from surreal_amd import _lib as L
K = HipKernels()
B, N, D, H1, H2 = 1024, 128, 376, 300, 200
g = torch.Generator(device='cuda').manual_seed(3)
obs = torch.randn(B, N, D, generator=g, device='cuda')
from surreal_amd.model.ppo_net import Mlp3Params
flat = (torch.rand(Mlp3Params.count(D, H1, H2, 1), generator=g, device='cuda') * 2 - 1) * 0.05
net = Mlp3Params(flat, 0, D, H1, H2, 1)
pd = torch.empty(K.mlp3_packed_numel(net), device='cuda')
K.mlp3_pack(net, pd)
zm = torch.randn(D, device='cuda') * 0.1
zs = torch.rand(D, device='cuda') + 0.5
out = torch.empty(B * N, device='cuda')
fn = lambda: K.mlp3_forward_fused(pd, net, obs, None, zm, zs, out, L.SMX_ACT_NONE)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n): fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n
fl = 2.0 * B * N * (D * H1 + H1 * H2 + H2)
print('fused critic (%s): %.4f ms  %.1f TFLOP/s  frac %.3f' % ('32-row' if os.environ.get('SMX_FUSED32') else '16-row',
      t, fl / t / 1e9, fl / t / 1e9 / 157.3))
if os.environ.get('SMX_FUSED_TBUF'):
    import ctypes, numpy as np
    lib = K.lib
    lib.smx_mlp3_fused_debug_tbuf.argtypes = [ctypes.c_void_p]
    lib.smx_mlp3_fused_debug_tbuf.restype = None
    tb = torch.zeros(B * N // 128, 16, dtype=torch.int64, device='cuda')
    lib.smx_mlp3_fused_debug_tbuf(ctypes.c_void_p(tb.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.smx_mlp3_fused_debug_tbuf(None)
    t = tb.cpu().numpy().astype(np.float64)
    t0 = t[:, 0].min()
    d = np.diff(t[:, :6], axis=1)
    print('phase cycles (median over WGs): prologue %.0f  layer1 %.0f  hand-over %.0f  layer2 %.0f  epilogue %.0f  total %.0f'
          % (*np.median(d, axis=0), np.median(t[:, 5] - t[:, 0])))
    print('kernel span %.0f cycles; WG start offsets quartiles %s' % (t[:, 5].max() - t0, np.percentile(t[:, 0] - t0, [0, 25, 50, 75, 100]).round()))
    print('chunk 5 of layer 1 (wave 0): ', np.median(np.diff(t[:, 8:15], axis=1), axis=0), ' = half0a half0b half1a stores barrier half1b')
    print('stores phase: wait-x %.0f  zfilter+x stores %.0f  W stores %.0f' % (np.median(t[:, 15] - t[:, 11]), np.median(t[:, 7] - t[:, 15]), np.median(t[:, 12] - t[:, 7])))
    order = np.argsort(t[:, 0])
    print('first starts', (t[order[:6], 0] - t0), 'ends of first WGs', (t[order[:6], 5] - t0))
    print('sum of WG totals / 256 CUs = %.0f cycles' % ((t[:, 5] - t[:, 0]).sum() / 256))
