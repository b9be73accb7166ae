"""LSTM kernels vs torch.nn.LSTM (CPU fp32) -- forward, final state, BPTT gradients"""
import sys, os, types, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from surreal_amd.kernels import HipKernels
from surreal_amd import _lib as L
K = HipKernels()
torch.manual_seed(0)

def run(B, T, D, H, with_cells=True):
    ref = torch.nn.LSTM(D, H, 1, batch_first=True)
    x = torch.randn(B, T, D)
    h0 = 0.3 * torch.randn(1, B, H); c0 = 0.3 * torch.randn(1, B, H)
    cells = (h0, c0) if with_cells else None
    out, (hN, cN) = ref(x, cells)
    dout = torch.randn(B, T, H)
    ref.zero_grad(); (out * dout).sum().backward()
    dev = 'cuda'
    n = L.load().smx_lstm_param_count(D, H)
    flat = torch.cat([p.detach().reshape(-1) for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]).to(dev)
    assert flat.numel() == n
    o = [0, 4*H*D, 4*H*D + 4*H*H, 4*H*D + 4*H*H + 4*H]
    net = types.SimpleNamespace(desc=L.Lstm(*(ctypes.c_void_p(flat[a:].data_ptr()) for a in o), D, H))
    xd = x.to(dev).contiguous()
    gates = torch.empty(B, T, 4*H, device=dev); od = torch.empty(B, T, H, device=dev); cs = torch.empty_like(od); hp = torch.empty_like(od)
    hNd = torch.empty(B, H, device=dev); cNd = torch.empty(B, H, device=dev)
    h0d = h0[0].to(dev).contiguous() if with_cells else None
    c0d = c0[0].to(dev).contiguous() if with_cells else None
    K.lstm_forward(net, xd, B, T, h0d, c0d, gates, od, cs, hp, hNd, cNd)
    torch.cuda.synchronize()
    e_out = (od.cpu() - out).abs().max().item()
    e_h = (hNd.cpu() - hN[0]).abs().max().item(); e_c = (cNd.cpu() - cN[0]).abs().max().item()
    grads = torch.zeros(n, device=dev); dg = torch.empty_like(gates)
    K.lstm_backward(net, xd, B, T, c0d, gates, cs, hp, dout.to(dev).contiguous(), dg, grads)
    torch.cuda.synchronize()
    gref = torch.cat([p.grad.reshape(-1) for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)])
    rel = ((grads.cpu() - gref).abs().max() / gref.abs().max()).item()
    print('B=%4d T=%3d D=%3d H=%3d cells=%d  out %.2e  hN %.2e cN %.2e  grad rel %.2e (|g|max %.2e)' % (B, T, D, H, with_cells, e_out, e_h, e_c, rel, gref.abs().max()))

for cfg in [(2, 21, 17, 100), (2, 26, 17, 100), (5, 4, 7, 12), (37, 9, 17, 100), (64, 21, 17, 100), (3, 5, 9, 128), (20, 6, 11, 256)]:
    run(*cfg)
run(4, 7, 17, 100, with_cells=False)
