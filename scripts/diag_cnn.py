"""CNN stem (im2col GEMMs) vs torch.nn.Conv2d net on CPU: forward features and all gradients"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn as nn
from surreal_amd.kernels import HipKernels
from surreal_amd.model.cnn_stem import CnnParams, CnnStem
K = HipKernels()
torch.manual_seed(0)

def run(F, C, H, W, feat, u8=True):
    ref = nn.Sequential(nn.Conv2d(C, 16, 8, 4), nn.ReLU(), nn.Conv2d(16, 32, 4, 2), nn.ReLU(), nn.Flatten())
    with torch.no_grad():
        n_flat = ref(torch.zeros(1, C, H, W)).shape[1]
    fc = nn.Linear(n_flat, feat)
    frames = torch.randint(0, 256, (F, C, H, W), dtype=torch.uint8)
    y = torch.relu(fc(ref(frames.float() / 255.0)))
    dy = torch.randn(F, feat)
    (y * dy).sum().backward()
    n = CnnParams.count((C, H, W), feat)
    flat = torch.zeros(n, device='cuda')
    p = CnnParams(flat, 0, (C, H, W), feat)
    src = {'conv1.W': ref[0].weight, 'conv1.b': ref[0].bias, 'conv2.W': ref[2].weight, 'conv2.b': ref[2].bias, 'fc.W': fc.weight, 'fc.b': fc.bias}
    for k, v in p.views.items(): v.copy_(src[k].detach())
    stem = CnnStem(K)
    ws = stem.workspace(p, F, 'cuda')
    D = 5
    xin = torch.zeros(F, D + feat, device='cuda')
    fr = frames.cuda() if u8 else frames.float().cuda()
    stem.forward(p, fr, F, ws, xin[:, D:])
    torch.cuda.synchronize()
    e_f = (xin[:, D:].cpu() - y.detach()).abs().max().item()
    dxin = torch.zeros_like(xin)
    dxin[:, D:] = (dy * (y.detach() > 0)).cuda()
    grads = torch.zeros(n, device='cuda')
    stem.backward(p, F, ws, dxin[:, D:], grads)
    torch.cuda.synchronize()
    gp = CnnParams(grads, 0, (C, H, W), feat)
    errs = []
    for k, v in gp.views.items():
        g = src[k].grad
        errs.append('%s %.1e' % (k, ((v.cpu() - g).abs().max() / (g.abs().max() + 1e-30)).item()))
    print('F=%4d C=%d %dx%d feat=%d u8=%d: feat err %.2e | grad rel: %s' % (F, C, H, W, feat, u8, e_f, ' '.join(errs)))

run(3, 3, 20, 20, 8)
run(5, 3, 36, 28, 24, u8=False)
run(4, 3, 84, 84, 256)
run(64, 3, 84, 84, 256)
