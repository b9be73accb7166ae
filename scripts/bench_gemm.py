"""micro-benchmark of the layer GEMM kernel (GPU box): time vs K and vs tile count"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from surreal_amd.kernels import HipKernels
K = HipKernels()

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

print('--- kc/kc  C[M,N] = A[M,K] . B[N,K]^T')
for M, N, Kd in [(1024, 320, 32), (1024, 320, 128), (1024, 320, 384), (1024, 320, 768), (1024, 320, 1536),
                 (320, 384, 1024), (320, 384, 1040), (32, 32, 1024), (32, 32, 4096), (320, 384, 256), (3200, 384, 256)]:
    A = torch.randn(M, Kd, device='cuda'); B = torch.randn(N, Kd, device='cuda'); C = torch.empty(M, N, device='cuda')
    t = timeit(lambda: K.linear(A, 1, B, 1, None, C, M, N, Kd))
    tiles = ((M + 31) // 32) * ((N + 31) // 32)
    print('M=%5d N=%4d K=%5d tiles=%5d  %7.2f us   (%.1f GFLOP/s-eq %.1f TF)' % (M, N, Kd, tiles, t, 0, 2.0 * M * N * Kd / t / 1e6))
print('--- non-kc dW-like: C[M,N] = A^T B, A [K,M], B [K,N]')
for M, N, Kd in [(320, 384, 256), (320, 384, 1024)]:
    A = torch.randn(Kd, M, device='cuda'); B = torch.randn(Kd, N, device='cuda'); C = torch.empty(M, N, device='cuda')
    t = timeit(lambda: K.linear(A, 0, B, 0, None, C, M, N, Kd))
    print('M=%5d N=%4d K=%5d  %7.2f us  %.1f TF' % (M, N, Kd, t, 2.0 * M * N * Kd / t / 1e6))
print('--- trivial kernel (fill 1 float)')
x = torch.zeros(4, device='cuda')
print('%.2f us' % timeit(lambda: K.fill(x, 1.0)))
