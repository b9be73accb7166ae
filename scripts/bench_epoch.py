"""micro-benchmark of the fused epoch kernels at the benchmark shape (GPU box): host-side launch times and,
with a timing build (SMX_EXTRA_FLAGS=-DSMX_EPOCH_TIMING python -m surreal_amd.build --force), the per-phase
cycle counts of every workgroup (thread 0) of the forward and the backward kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from surreal_amd import _lib as L
from surreal_amd.kernels import HipKernels
import test_gpu_epoch as TE
K = HipKernels()


def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows, D, H1, H2, A = 1024, 376, 300, 200, 17
T = TE.build(rows, D, H1, H2, A, seed=1, mode=L.SMX_PPO_ADAPT, device='cuda')
t = T['d']
t['ctrl'][L.C_KL_TARGET] = 1e9
f = timeit(lambda: TE.run(K, t, L.SMX_PPO_ADAPT, phase='fwd'))
b = timeit(lambda: TE.run(K, t, L.SMX_PPO_ADAPT, phase='bwd'))
print('eager launches (host-bound): pack + forward %.2f us   backward + wgrad %.2f us' % (f, b))

# ---- per-phase timestamps of the forward kernel (thread 0 of every workgroup) ----
import ctypes
lib = K.lib
lib.smx_epoch_debug_tbuf.argtypes = [ctypes.c_void_p]
lib.smx_epoch_debug_tbuf.restype = None
tb = torch.zeros(512 * 32, dtype=torch.int64, device='cuda')
lib.smx_epoch_debug_tbuf(ctypes.c_void_p(tb.data_ptr()))
for _once in (0,):
  for _ in range(3):
      TE.run(K, t, L.SMX_PPO_ADAPT, phase='fwd')
  torch.cuda.synchronize()
  TT = tb.view(512, 32)[:128].cpu().double()
  T_ = TT[:, :7]
  for l in range(3):
      st = TT[:, 2 + l]
      print('  layer %d: setup %.0f  main loop %.0f  epilogue %.0f  barrier wait %.0f' % (l + 1, (TT[:, 16 + 4 * l] - st).mean(), (TT[:, 17 + 4 * l] - TT[:, 16 + 4 * l]).mean(), (TT[:, 18 + 4 * l] - TT[:, 17 + 4 * l]).mean(), (TT[:, 3 + l] - TT[:, 18 + 4 * l]).mean()))
  base = T_[:, 0].min()
  print('start spread (cycles): max-min of stamp0 = %.0f' % (T_[:, 0].max() - base))
  # 
  names = ['prologue loads+LDS', 'barrier', 'layer1', 'layer2', 'layer3', 'loss']
  for i, nm in enumerate(names):
      d = T_[:, i + 1] - T_[:, i]
      print('%-20s mean %7.0f  min %7.0f  max %7.0f cycles   (actor wgs %.0f, critic wgs %.0f)' % (
          nm, d.mean(), d.min(), d.max(), d[:64].mean(), d[64:].mean()))
  tot = T_[:, 6] - T_[:, 0]
  print('total in-kernel: mean %.0f max %.0f ; end-start over all wgs %.0f cycles' % (tot.mean(), tot.max(), T_[:, 6].max() - base))
tb.zero_()
for _ in range(3):
    TE.run(K, t, L.SMX_PPO_ADAPT, phase='bwd')
torch.cuda.synchronize()
TT = tb.view(512, 32)[:256].cpu().double()
pol = TT[:, 2] > 0
for nm, sel in (('actor', pol), ('critic', ~pol)):
    X = TT[sel]
    print('bwd %s wgs (%d): entry->flag %.0f | partials+scalars %.0f | dz3 tile+barrier %.0f' % (
        nm, X.shape[0], (X[:, 1] - X[:, 0]).mean(), ((X[:, 2] - X[:, 1]).mean() if nm == 'actor' else 0),
        ((X[:, 3] - X[:, 2]).mean() if nm == 'actor' else (X[:, 3] - X[:, 1]).mean())))
    for l in range(2):
        st = X[:, 3 + l]
        print('   layer dz%d: setup %.0f main %.0f epilogue %.0f barrier %.0f' % (2 - l, (X[:, 8 + 4 * l] - st).mean(),
              (X[:, 9 + 4 * l] - X[:, 8 + 4 * l]).mean(), (X[:, 10 + 4 * l] - X[:, 9 + 4 * l]).mean(),
              (X[:, 4 + l] - X[:, 10 + 4 * l]).mean()))
    print('   total %.0f (max %.0f)' % ((X[:, 5] - X[:, 0]).mean(), (X[:, 5] - X[:, 0]).max()))
lib.smx_epoch_debug_tbuf(None)
