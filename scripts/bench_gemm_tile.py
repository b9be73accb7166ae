"""time smx_linear_f32 on the stems' big shapes (tests/diag/gemm_tile_cases.py); run it with SMX_GEMM_ROWS_ONLY=1 for
gemm_rows_kernel (the A/B switch of csrc/smx_gemm.hip::launch_tiles)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'diag'))
import torch
import gemm_tile_cases as GC
from surreal_amd import kernels as KN
K = KN.default_kernels()
tag = 'rows' if os.environ.get('SMX_GEMM_ROWS_ONLY') else 'tile'
for ci, (M, N, Kd, akc, bkc, bias, act, mask) in enumerate(GC.CASES):
    g = torch.Generator(device='cuda').manual_seed(ci)
    A = torch.randn((M, Kd) if akc else (Kd, M), device='cuda', generator=g)
    B = torch.randn((N, Kd) if bkc else (Kd, N), device='cuda', generator=g)
    b = torch.randn(N, device='cuda') if bias else None
    mk = torch.ones(M, N, device='cuda') if mask else None
    C = torch.empty(M, N, device='cuda')
    for _ in range(3):
        K.linear(A, akc, B, bkc, b, C, M, N, Kd, act=act, relu_mask=mk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.linear(A, akc, B, bkc, b, C, M, N, Kd, act=act, relu_mask=mk)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('%-6s M=%6d N=%4d K=%4d kc=%d%d: %7.1f us  %6.1f TFLOP/s' % (tag, M, N, Kd, akc, bkc, us, 2.0 * M * N * Kd / us / 1e6), flush=True)
