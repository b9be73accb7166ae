#!/usr/bin/env python
"""The HBM-bound launches of the path, ten times each, at their benchmark shapes -- to be run under rocprofv3
(`--kernel-trace --stats`, then `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE passes) so that
scripts/profiles_hbm_small.py can put GB/s against the 8 TB/s peak next to each: gae_norm, window_emit, ring_insert,
gather_rows (FIFO pop), uniform_gather_multi (512 of 1e6 rows, five fields), zupdate, clip_adam, reward_filter.
Prints the ALGORITHMIC bytes per launch of each (what must be read + written once) as one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from surreal_amd import kernels as KN, _lib as L  # noqa: E402

REPS = 10


def main():
    K = KN.default_kernels()
    dev = 'cuda'
    f = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    alg = {}
    B, N, D, A = 1024, 128, 376, 17
    # gae_norm: values + rewards + dones in, adv + ret out (ppo.py:387-418)
    vals, tail, rew = f(B * N), f(B), f(B, N)
    dones = (torch.rand(B, N, device=dev) < 0.01).float()
    idx = torch.arange(N, dtype=torch.float32)
    gpow, lpow = torch.pow(0.995, idx).to(dev), torch.pow(0.97, idx).to(dev)
    adv, ret, mom, ticket = f(B), f(B), f(3), torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(REPS):
        K.gae_norm(vals, rew, dones, gpow, lpow, 0.995, 0.995 ** N, B, N, N, adv, ret, mom, 1e-4, ticket, values_tail=tail)
    alg['gae_norm_kernel'] = 4.0 * (3 * B * N + B + 2 * B)
    # window cut / FIFO insert / FIFO pop of the observation field (197 MB each way)
    roll, win = f(B, N + 1, D), torch.empty(B, N, D, device=dev)
    cap = 2 * B + 3
    table, dst = torch.empty(cap, N * D, device=dev), torch.empty(B, N * D, device=dev)
    ix = (torch.arange(B, device=dev) + 700) % cap
    for _ in range(REPS):
        K.window_emit(roll, 0, N, N, 1, win)
        K.ring_insert(table, B + 5, win.view(B, N * D))
        K.gather_rows(table, ix, dst)
    alg['window_emit_kernel'] = alg['ring_insert_kernel'] = alg['gather_rows_kernel'] = 2.0 * 4 * B * N * D
    # uniform sample: 512 of 1e6 SSAR rows, five fields, one launch (uniform_replay.py:36-47)
    cap3, B3 = 1000000, 512
    widths = (17, 17, 6, 1, 1)
    tabs = [f(cap3, w) for w in widths]
    outs = [torch.empty(B3, w, device=dev) for w in widths]
    for k in range(REPS):
        K.uniform_gather_multi(tabs, outs, cap3, 7, k * B3)
    alg['uniform_gather_multi_kernel'] = 2.0 * 4 * B3 * sum(widths)
    # z-filter update over the step-0 observations (z_filter.py:44-59) and over an LSTM learn's B x E rows
    for rows, d in ((B, D), (7936, 17)):
        x, rs, rq, cnt = f(rows, d), torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.ones(1, device=dev)
        for _ in range(REPS):
            K.zfilter_update(x, rs, rq, cnt, rows)
        alg['zupdate_kernel@%dx%d' % (rows, d)] = 4.0 * rows * d
    # clip_grad_norm_ + Adam of both groups + the packed copies (ppo.py:243-247)
    from surreal_amd.model.ppo_net import Mlp3Params
    nets = []
    for out_dim in (A, 1):
        flat = (torch.rand(Mlp3Params.count(D, 300, 200, out_dim), device=dev) * 2 - 1) * 0.05
        net = Mlp3Params(flat, 0, D, 300, 200, out_dim)
        net.flat = flat
        nets.append(net)
    act, cri = nets
    ctrl = torch.zeros(L.CTRL_WORDS, device=dev)
    ctrl[L.C_LR_ACTOR], ctrl[L.C_LR_CRITIC], ctrl[L.C_ACTOR_MAX_NORM], ctrl[L.C_CRITIC_MAX_NORM] = 1e-4, 1e-4, 5.0, 5.0
    ctrl.view(torch.int32)[L.C_STEP_ACTOR:L.C_STEP_CRITIC + 1] = 1
    grp = []
    for net in (act, cri):
        n = net.flat.numel()
        grp.append((net.flat, f(n) * 1e-3, torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.rand(512, device=dev) * 1e-6,
                    400, False, torch.zeros(1, device=dev)))
    pk = [(net, torch.zeros(K.epoch_packed_numel(net), device=dev)) for net in (act, cri)]
    for _ in range(REPS):
        K.clip_adam_pair(grp[0], grp[1], ctrl, pack=tuple(pk))
    n_par = act.flat.numel() + cri.flat.numel()
    alg['clip_adam_kernel'] = 4.0 * (4 * n_par + 3 * n_par) + 4.0 * sum(p.numel() for _, p in pk)
    # reward scale + RewardFilter (ppo.py:452-455)
    state = torch.tensor([1000.0, 10.0, 1200.0], device=dev)
    part = torch.zeros(K.reward_filter_partials(), device=dev, dtype=torch.float64)
    out = torch.empty(B, N, device=dev)
    for _ in range(REPS):
        K.reward_filter(rew, 0.5, state, 1e-5, out, part, ticket)
    alg['reward_filter_kernel'] = 2.0 * 4 * B * N
    torch.cuda.synchronize()
    print(json.dumps({'algorithmic_bytes_per_launch': alg, 'reps': REPS}))


if __name__ == '__main__':
    main()
