"""Shapes of the stems' big dense layers run through smx_linear_f32; used twice by
tests/test_gpu_kernels.py::test_linear_tile_kernel_*: in-process (the LDS-tiled gemm_tile_kernel takes them) and in a
subprocess with SMX_GEMM_ROWS_ONLY=1 (gemm_rows_kernel takes them) -- the outputs must be the same BITS.

    python tests/diag/gemm_tile_cases.py out.npz      # writes every case's output
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# (M, N, K, a_kc, b_kc, bias, act, mask): a_kc / b_kc = 1 K-contiguous, 0 K-strided
CASES = [
    (7168, 256, 2592, 1, 1, True, 1, False),       # the Linear behind the convolutions, forward (configs[3])
    (7168, 2592, 256, 1, 0, False, 0, True),       # its data gradient (+ the ReLU mask of the conv output)
    (7936, 300, 100, 1, 1, True, 1, False),        # LSTM output -> first MLP layer over B*T rows (configs[1])
    (7936, 100, 300, 1, 0, False, 0, False),       # ... and the gradient back into the LSTM output
    (2049, 65, 36, 1, 1, True, 2, False),          # ragged everything, tanh; 64 x 64 tiles
    (4100, 130, 68, 0, 1, False, 0, False),        # K-strided A
    (2304, 200, 132, 0, 0, True, 1, True),         # both K-strided
    (131072, 300, 100, 1, 1, True, 1, False),      # B = 1024 x 128 rows
]


def run_cases(kern, device='cuda'):
    from surreal_amd import _lib as L
    outs = []
    for ci, (M, N, Kd, akc, bkc, bias, act, mask) in enumerate(CASES):
        g = torch.Generator(device=device).manual_seed(100 + ci)
        A = torch.randn((M, Kd) if akc else (Kd, M), device=device, generator=g)
        B = torch.randn((N, Kd) if bkc else (Kd, N), device=device, generator=g) / (Kd ** 0.5)
        b = torch.randn(N, device=device, generator=g) if bias else None
        mk = (torch.rand(M, N, device=device, generator=g) > 0.4).float() if mask else None
        C = torch.full((M, N), float('nan'), device=device)
        kern.linear(A, akc, B, bkc, b, C, M, N, Kd, act=act, relu_mask=mk)
        outs.append((A, B, b, mk, C))
    torch.cuda.synchronize()
    return outs


if __name__ == '__main__':
    from surreal_amd import kernels as KN
    outs = run_cases(KN.default_kernels())
    np.savez(sys.argv[1], **{'c%d' % i: o[-1].cpu().numpy() for i, o in enumerate(outs)})
