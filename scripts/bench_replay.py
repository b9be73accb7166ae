#!/usr/bin/env python
"""Replay / windowing kernels against the HBM roofline: bytes
moved per launch / launch time, HIP events on the launch stream, 20 launches each.

  window_emit   the moving-window cut of a rollout (exp_sender_wrapper.py:209-228), cfg 5 shape
  ring_insert   FIFO / uniform insert of the cut windows (fifo_replay.py:27, uniform_replay.py:31-46)
  gather_rows   FIFO pop of one learner batch; uniform sample of 512 SSARs out of 1e6 (cfg 3)
  uniform_indices   512 draws with replacement (uniform_replay.py:48-56)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surreal_amd import kernels as KN  # noqa: E402

PEAK = 8000.0


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    K = KN.default_kernels()
    dev = 'cuda'
    out = {}
    n, T, D = 1024, 128, 376
    roll = torch.randn(n, T + 1, D, device=dev)
    win = torch.empty(n, T, D, device=dev)
    t = timed(lambda: K.window_emit(roll, 0, T, T, 1, win))
    by = 2.0 * win.numel() * 4
    out['window_emit obs 1024x128x376'] = {'us': t * 1e6, 'GBps': by / t / 1e9, 'frac_hbm': by / t / 1e9 / PEAK}
    cap = 2 * n + 3
    table = torch.empty(cap, T * D, device=dev)
    t = timed(lambda: K.ring_insert(table, n + 5, win.view(n, T * D)))
    out['ring_insert 1024 rows x 192 KB'] = {'us': t * 1e6, 'GBps': by / t / 1e9, 'frac_hbm': by / t / 1e9 / PEAK}
    idx = (torch.arange(n, device=dev) + 700) % cap
    dst = torch.empty(n, T * D, device=dev)
    t = timed(lambda: K.gather_rows(table, idx, dst))
    out['gather_rows FIFO pop 1024 rows x 192 KB'] = {'us': t * 1e6, 'GBps': by / t / 1e9, 'frac_hbm': by / t / 1e9 / PEAK}
    # a plain device copy of the same bytes, for scale
    t = timed(lambda: dst.copy_(win.view(n, T * D)))
    out['torch copy_ of the same 197 MB'] = {'us': t * 1e6, 'GBps': by / t / 1e9, 'frac_hbm': by / t / 1e9 / PEAK}
    # cfg 3: uniform replay of 1e6 SSAR rows (obs 17 | obs_next 17 | action 6 | reward | done)
    cap3, B3 = 1000000, 512
    for name, width in (('obs', 17), ('actions', 6), ('rewards', 1)):
        tab = torch.randn(cap3, width, device=dev)
        ix = torch.empty(B3, dtype=torch.int64, device=dev)
        K.uniform_indices(ix, cap3, 7, 0)
        d3 = torch.empty(B3, width, device=dev)
        t = timed(lambda: K.gather_rows(tab, ix, d3))
        out['gather_rows uniform 512 of 1e6 x %d floats' % width] = {'us': t * 1e6, 'GBps': 2.0 * B3 * width * 4 / t / 1e9}
    t = timed(lambda: K.uniform_indices(ix, cap3, 7, 0))
    out['uniform_indices 512'] = {'us': t * 1e6}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
