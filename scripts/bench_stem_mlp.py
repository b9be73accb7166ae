"""the MLP on top of a stem over B*E rows (126 976 x 100 -> 300 -> 200 -> A at the 1024 x 128 LSTM shapes): forward
layered vs fused (K.mlp3_forward(pack=...)), backward (data gradients + split-K weight gradients; SMX_WGRAD_TILED=1 in the
environment keeps the weight gradients on the tiled GEMM for the A/B)
    python scripts/bench_stem_mlp.py [rows D A]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from surreal_amd.kernels import HipKernels
from surreal_amd import _lib as L
from surreal_amd.model.ppo_net import Mlp3Params

rows, D, A = [int(v) for v in sys.argv[1:4]] or [126976, 100, 6]
H1, H2 = 300, 200
K = HipKernels()
g = torch.Generator(device='cuda').manual_seed(3)
flat = (torch.rand(Mlp3Params.count(D, H1, H2, A), generator=g, device='cuda') * 2 - 1) * 0.05
net = Mlp3Params(flat, 0, D, H1, H2, A)
x = torch.randn(rows, D, generator=g, device='cuda')
f = lambda *s: torch.empty(*s, device='cuda')  # noqa: E731
h1, h2, out = f(rows, H1), f(rows, H2), f(rows, A)
dz3, dz2, dz1 = torch.randn(rows, A, generator=g, device='cuda') / rows, f(rows, H2), f(rows, H1)
grads = f(net.numel)
pack = f(K.mlp3_packed_numel(net))
ws = f(max(K.mlp3_backward_ws_floats(net, rows), 1))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fl = 2.0 * rows * (D * H1 + H1 * H2 + H2 * A)
t = timed(lambda: K.mlp3_forward(net, x, h1, h2, out, L.SMX_ACT_TANH))
print('forward layered: %8.1f us  %6.1f TFLOP/s' % (t, fl / t / 1e6))
t = timed(lambda: K.mlp3_forward(net, x, h1, h2, out, L.SMX_ACT_TANH, pack=pack))
print('forward fused  : %8.1f us  %6.1f TFLOP/s' % (t, fl / t / 1e6))
t = timed(lambda: K.mlp3_backward(net, x, h1, h2, dz3, dz2, dz1, grads, None, ws=ws))
print('backward layered dgrad + split-K wgrad%s: %8.1f us  %6.1f TFLOP/s' % (
    ' (tiled)' if os.environ.get('SMX_WGRAD_TILED') else '', t, (2 * fl - 2.0 * rows * D * H1) / t / 1e6))
npt = K.mlp3_dgrad_rows_ws_floats(net)
if npt:
    packT, dx = f(npt), f(rows, D)
    t = timed(lambda: K.mlp3_backward(net, x, h1, h2, dz3, dz2, dz1, grads, None, ws=ws, packT=packT, dx=dx))
    print('backward fused dgrad (+ dx) + wgrad    : %8.1f us  %6.1f TFLOP/s' % (t, 2 * fl / t / 1e6))
