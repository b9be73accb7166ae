"""GPU tier (-m gpu): every HIP entry point, called through the C ABI, against the torch-CPU
statement of the same contract (tests/cpu_kernels.py) on identical seeded inputs.
Tolerance: 1e-5 absolute + 1e-5 relative (fp32), the north-star bound."""
import os
import types

import numpy as np
import pytest
import torch

from surreal_amd import _lib as L
from cpu_kernels import TorchCpuKernels

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-5, 1e-5


@pytest.fixture(scope='module')
def K():
    from surreal_amd.kernels import HipKernels
    return HipKernels()


C = TorchCpuKernels()


def dev(t):
    return t.cuda() if torch.is_tensor(t) else t


def close(a, b, atol=ATOL, rtol=RTOL, msg=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=msg)


def make_net(D, H1, H2, OUT, seed, device):
    from surreal_amd.model.ppo_net import Mlp3Params
    g = torch.Generator().manual_seed(seed)
    n = Mlp3Params.count(D, H1, H2, OUT)
    flat = (torch.rand(n, generator=g) * 2 - 1)
    net_c = Mlp3Params(flat.clone(), 0, D, H1, H2, OUT)
    for nm, v in net_c.views.items():
        fan = net_c.views['W' + nm[1]].shape[1]
        v.mul_(1.0 / np.sqrt(fan))
    flat_d = torch.cat([v.reshape(-1) for v in net_c.views.values()]).to(device)
    net_d = Mlp3Params(flat_d, 0, D, H1, H2, OUT)
    return net_c, net_d


# ------------------------------------------------------------------------------------------
def test_zfilter(K):
    g = torch.Generator().manual_seed(0)
    for D, rows, T in ((17, 64, 5), (376, 1000, 3), (5, 7, 1)):
        x3 = torch.randn(rows, T, D, generator=g) * 3 + 0.5
        rs = torch.randn(D, generator=g) * 100
        rsq = torch.rand(D, generator=g) * 5000 + 300
        cnt = torch.tensor([1000.0 + 1e-5])
        mc, sc = torch.empty(D), torch.empty(D)
        md, sd = torch.empty(D).cuda(), torch.empty(D).cuda()
        C.zfilter_stats(rs, rsq, cnt, 1e-5, mc, sc)
        K.zfilter_stats(dev(rs), dev(rsq), dev(cnt), 1e-5, md, sd)
        close(md, mc), close(sd, sc)
        xv_c, x3d = x3[:, 0, :], x3.cuda()
        oc, od = torch.empty(rows, D), torch.empty(rows, D).cuda()
        C.zfilter_forward(xv_c, mc, sc, oc)
        K.zfilter_forward(x3d[:, 0, :], md, sd, od)          # strided view: ldx = T*D
        close(od, oc, msg='zforward D=%d' % D)
        rs_d, rsq_d, cnt_d = dev(rs.clone()), dev(rsq.clone()), dev(cnt.clone())
        C.zfilter_update(xv_c, rs, rsq, cnt, rows)
        K.zfilter_update(x3d[:, 0, :], rs_d, rsq_d, cnt_d, rows)
        close(rs_d, rs, rtol=1e-5, atol=1e-3), close(rsq_d, rsq, rtol=1e-5, atol=1e-2), close(cnt_d, cnt)


def test_zfilter_against_the_reference_restatement(K):
    """statistics, forward (reciprocal: <= 1 ulp from the division) and update against oracle/ppo_oracle.py's ZFilter
    (surreal/model/z_filter.py:40-79) in float64"""
    import ppo_oracle
    g = torch.Generator().manual_seed(21)
    for D, rows in ((17, 64), (376, 1000), (5, 7)):
        x = torch.randn(rows, D, generator=g) * 3 + 0.5
        cnt = 1000.0 + 1e-5
        rs = torch.randn(D, generator=g) * 100
        rsq = rs * rs / cnt + (torch.rand(D, generator=g) * 4 + 0.5) * cnt       # a valid state: variance 0.5 .. 4.5
        state = dict(running_sum=rs.numpy(), running_sumsq=rsq.numpy(), count=np.array([cnt], dtype=np.float32))
        md, sd, od = torch.empty(D).cuda(), torch.empty(D).cuda(), torch.empty(rows, D).cuda()
        rs_d, rsq_d, cnt_d = dev(rs.clone()), dev(rsq.clone()), dev(torch.tensor([cnt]))
        K.zfilter_stats(rs_d, rsq_d, cnt_d, 1e-5, md, sd)
        K.zfilter_forward(dev(x), md, sd, od)
        K.zfilter_update(dev(x), rs_d, rsq_d, cnt_d, rows)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            Z = ppo_oracle.ZFilter(D, state=state)
            want = Z.forward(x.double())
            Z.z_update(x.double())
        finally:
            torch.set_default_dtype(prev)
        close(od.cpu().double(), want, atol=5e-6, rtol=1e-5, msg='z-filtered rows vs float64, D=%d' % D)
        close(rs_d.cpu().double(), Z.running_sum, rtol=1e-5, atol=1e-3, msg='running_sum')
        close(rsq_d.cpu().double(), Z.running_sumsq, rtol=1e-5, atol=1e-2, msg='running_sumsq')
        close(cnt_d.cpu().double(), Z.count, rtol=1e-6, atol=0, msg='count')


@pytest.mark.parametrize('B,N,H', [(8, 12, 12), (37, 19, 19), (1024, 128, 128), (5, 25, 5),
                                   (64, 128, 5), (3, 200, 1), (2, 1, 1)])
def test_windowed_gae(K, B, N, H):
    g = torch.Generator().manual_seed(B * 1000 + N)
    values = torch.randn(B * (N + 1), generator=g) * 3
    rewards = torch.randn(B, N, generator=g)
    dones = (torch.rand(B, N, generator=g) < 0.1).float()
    idx = torch.tensor(range(N), dtype=torch.float32)
    gpow, lpow = torch.pow(0.995, idx), torch.pow(0.97, idx)
    E = N - H + 1
    ac, rc = torch.empty(B * E), torch.empty(B * E)
    ad, rd = torch.empty(B * E).cuda(), torch.empty(B * E).cuda()
    C.gae(values, rewards, dones, gpow, lpow, 0.995, 0.995 ** H, B, N, H, ac, rc)
    K.gae(dev(values), dev(rewards), dev(dones), dev(gpow), dev(lpow), 0.995, 0.995 ** H, B, N, H, ad, rd)
    close(ad, ac, msg='adv'), close(rd, rc, msg='ret')
    # split layout: values [B, N] + values_tail [B]
    v2 = values.view(B, N + 1)
    K.gae(dev(v2[:, :N].contiguous()), dev(rewards), dev(dones), dev(gpow), dev(lpow), 0.995,
          0.995 ** H, B, N, H, ad, rd, values_tail=dev(v2[:, N].contiguous()))
    close(ad, ac, msg='adv (split values)'), close(rd, rc, msg='ret (split values)')
    # size-independent property: linearity in the rewards (values = 0, no dones) and the
    # telescoping identity adv = ret - V_0 when lambda = 1
    one = torch.ones(N)
    zeros = torch.zeros(B * (N + 1)).cuda()
    if H == N:
        K.gae(dev(values), dev(rewards), torch.zeros(B, N).cuda(), dev(gpow), dev(one), 0.995,
              0.995 ** N, B, N, N, ad, rd)
        v0 = values.view(B, N + 1)[:, 0]
        close(ad.cpu(), rd.cpu() - v0, atol=2e-5, rtol=1e-5, msg='lambda=1 telescoping')


@pytest.mark.parametrize('B,N,H,rnn', [(8, 12, 12, False), (1024, 128, 128, False), (37, 19, 19, False), (5, 25, 5, True),
                                       (64, 128, 5, True), (3, 40, 1, True)])
def test_windowed_gae_against_the_reference_restatement(K, B, N, H, rnn):
    """the same launch against oracle/ppo_oracle.py's gae_from_values (surreal/learner/ppo.py:389-418, both the
    whole-trajectory rule and the LSTM policy's sliding windows of `horizon` steps) evaluated in float64 -- an oracle that
    shares no code with the kernel's CPU double"""
    import ppo_oracle
    g = torch.Generator().manual_seed(B * 7 + N)
    values = torch.randn(B, N + 1, generator=g) * 3
    rewards = torch.randn(B, N, generator=g)
    dones = (torch.rand(B, N, generator=g) < 0.1).float()
    idx = torch.tensor(range(N), dtype=torch.float32)
    gpow, lpow = torch.pow(0.995, idx), torch.pow(0.97, idx)
    E = N - H + 1
    ad, rd = torch.empty(B * E).cuda(), torch.empty(B * E).cuda()
    K.gae(dev(values.reshape(-1)), dev(rewards), dev(dones), dev(gpow), dev(lpow), 0.995, 0.995 ** H, B, N, H, ad, rd)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        O = ppo_oracle.OraclePPOLearner.__new__(ppo_oracle.OraclePPOLearner)     # the arithmetic only: no model
        O.gamma, O.lam, O.n_step, O.horizon, O.if_rnn_policy, O.norm_adv, O.batch_size = 0.995, 0.97, N, H, rnn, False, B
        masked = values.double().clone()
        masked[:, 1:] *= 1 - dones.double()                  # ppo.py:386
        adv, ret = O.gae_from_values(masked, rewards.double())
    finally:
        torch.set_default_dtype(prev)
    close(ad.cpu().double(), adv.reshape(-1), atol=2e-5, rtol=2e-5, msg='advantages vs float64')
    close(rd.cpu().double(), ret.reshape(-1), atol=2e-5, rtol=2e-5, msg='returns vs float64')


def test_moments_and_normalize(K):
    g = torch.Generator().manual_seed(3)
    for n in (2, 37, 1024, 127000):
        x = torch.randn(n, generator=g) * 2.5 + 7.0
        mc, md = torch.empty(3), torch.empty(3).cuda()
        C.moments(x, mc)
        K.moments(dev(x), md)
        close(md, mc, rtol=1e-6)
        np.testing.assert_allclose(float(torch.sqrt(md[2] / (n - 1))), float(x.std()), rtol=2e-6)
        xc, xd = x.clone(), dev(x.clone())
        C.adv_normalize(xc, mc, 1e-4)
        K.adv_normalize(xd, md, 1e-4)
        close(xd, xc)
        close(xd.cpu(), (x - x.mean()) / max(x.std(), 1e-4), atol=2e-5)
    # tiny spread -> the 1e-4 floor is used (ppo.py:405)
    x = torch.full((16,), 3.0) + torch.arange(16) * 1e-7
    md = torch.empty(3).cuda()
    K.moments(dev(x), md)
    xd = dev(x.clone())
    K.adv_normalize(xd, md, 1e-4)
    close(xd.cpu(), (x - x.double().mean().float()) / 1e-4, atol=1e-4)
    # merge of per-rank moments == moments of the concatenation
    a, b = torch.randn(300, generator=g), torch.randn(77, generator=g) * 4 + 1
    parts = torch.empty(2, 3).cuda()
    K.moments(dev(a), parts[0]), K.moments(dev(b), parts[1])
    out, full = torch.empty(3).cuda(), torch.empty(3).cuda()
    K.moments_merge(parts, out)
    K.moments(dev(torch.cat([a, b])), full)
    close(out, full, rtol=2e-6)


@pytest.mark.parametrize('rows,D,H1,H2,OUT,act', [
    (8, 11, 24, 16, 3, L.SMX_ACT_TANH), (37, 29, 40, 24, 5, L.SMX_ACT_TANH),
    (64, 17, 300, 200, 6, L.SMX_ACT_TANH), (1024, 376, 300, 200, 17, L.SMX_ACT_TANH),
    (1024, 376, 300, 200, 1, L.SMX_ACT_NONE), (33, 100, 300, 200, 1, L.SMX_ACT_NONE)])
def test_mlp3_forward_backward(K, rows, D, H1, H2, OUT, act):
    nc, nd = make_net(D, H1, H2, OUT, rows + D, 'cuda')
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g)
    mk = lambda *s: (torch.empty(*s), torch.empty(*s).cuda())  # noqa: E731
    (h1c, h1d), (h2c, h2d), (oc, od) = mk(rows, H1), mk(rows, H2), mk(rows, OUT)
    C.mlp3_forward(nc, x, h1c, h2c, oc, act)
    K.mlp3_forward(nd, dev(x), h1d, h2d, od, act)
    close(h1d, h1c, msg='h1'), close(h2d, h2c, msg='h2'), close(od, oc, msg='out')
    # A = I style transpose detector: out rows must follow x rows, not mix them
    dz3 = torch.randn(rows, OUT, generator=g) / rows
    (dz2c, dz2d), (dz1c, dz1d) = mk(rows, H2), mk(rows, H1)
    n = nc.numel
    gc, gd = torch.zeros(n), torch.zeros(n).cuda()
    npart = K.mlp3_backward_partials(nd)
    assert npart == C.mlp3_backward_partials(nc)
    sc, sd = torch.zeros(npart), torch.zeros(npart).cuda()
    C.mlp3_backward(nc, x, h1c, h2c, dz3, dz2c, dz1c, gc, sc)
    K.mlp3_backward(nd, dev(x), h1d, h2d, dev(dz3), dz2d, dz1d, gd, sd)
    close(dz2d, dz2c, msg='dz2'), close(dz1d, dz1c, msg='dz1')
    close(gd, gc, atol=2e-6, rtol=2e-5, msg='grads')
    np.testing.assert_allclose(float(sd.sum()), float(sc.sum()), rtol=1e-5)
    # stop flag turns both into no-ops
    stop = torch.ones(1, dtype=torch.int32).cuda()
    od.fill_(7.0)
    K.mlp3_forward(nd, dev(x), h1d, h2d, od, act, stop)
    assert float(od.min()) == 7.0


@pytest.mark.parametrize('rows,D,H1,H2,OUT,act', [
    (30000, 100, 300, 200, 6, L.SMX_ACT_TANH), (24576 + 77, 288, 300, 200, 8, L.SMX_ACT_TANH),
    (126976, 100, 300, 200, 17, L.SMX_ACT_TANH), (25000, 100, 300, 200, 1, L.SMX_ACT_NONE),
    (26000, 36, 320, 224, 32, L.SMX_ACT_NONE), (25001, 100, 128, 96, 3, L.SMX_ACT_TANH)])
def test_mlp3_forward_fused_over_many_rows_keeps_activations(K, rows, D, H1, H2, OUT, act):
    """K.mlp3_forward(pack=...) from FUSED_ROWS_MIN rows on: ONE launch of the 16-row fused kernel that also stores h1 / h2
    (smx_mlp3_forward_rows_f32) against the layered launches (smx_mlp3_forward_f32) and the torch-CPU statement; the
    stop flag; the strided output"""
    nc, nd = make_net(D, H1, H2, OUT, rows + D, 'cuda')
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g)
    xd = dev(x)
    mk = lambda *s: torch.full(s, 9.0, device='cuda')  # noqa: E731
    h1f, h2f, of = mk(rows, H1), mk(rows, H2), mk(rows, OUT)
    h1l, h2l, ol = mk(rows, H1), mk(rows, H2), mk(rows, OUT)
    pack = torch.empty(K.mlp3_packed_numel(nd), device='cuda')
    assert rows >= K.FUSED_ROWS_MIN and K.lib.smx_mlp3_forward_rows_supported(D, H1, H2, OUT)
    K.mlp3_forward(nd, xd, h1f, h2f, of, act, pack=pack)
    K.mlp3_forward(nd, xd, h1l, h2l, ol, act)
    close(h1f, h1l, msg='h1 vs layered'), close(h2f, h2l, msg='h2 vs layered'), close(of, ol, msg='out vs layered')
    n = min(rows, 4096)                     # the CPU statement on a slice (first and last rows)
    for sl in (slice(0, n), slice(rows - n, rows)):
        h1c, h2c, oc = torch.empty(n, H1), torch.empty(n, H2), torch.empty(n, OUT)
        C.mlp3_forward(nc, x[sl], h1c, h2c, oc, act)
        close(h1f[sl], h1c, msg='h1'), close(h2f[sl], h2c, msg='h2'), close(of[sl], oc, msg='out')
    # a second call after the weights moved (the pack is refreshed inside the call)
    for v in nd.views.values():
        v.mul_(0.5)
    K.mlp3_forward(nd, xd, h1f, h2f, of, act, pack=pack)
    K.mlp3_forward(nd, xd, h1l, h2l, ol, act)
    close(of, ol, msg='out after a weight change'), close(h2f, h2l, msg='h2 after a weight change')
    stop = torch.ones(1, dtype=torch.int32).cuda()
    of.fill_(7.0)
    K.mlp3_forward(nd, xd, h1f, h2f, of, act, stop, pack=pack)
    assert float(of.min()) == 7.0 and float(of.max()) == 7.0
    # output rows with a stride (a column block of a wider table)
    wide = torch.full((rows, OUT + 3), 5.0, device='cuda')
    K.mlp3_forward(nd, xd, h1f, h2f, wide[:, :OUT], act, pack=pack)
    close(wide[:, :OUT], ol, msg='strided out')
    assert float(wide[:, OUT:].min()) == 5.0 and float(wide[:, OUT:].max()) == 5.0


@pytest.mark.parametrize('rows,D,H1,H2,OUT', [(7936, 100, 300, 200, 6), (2100, 17, 64, 40, 1), (40000, 132, 300, 200, 12),
                                              (900, 100, 300, 200, 6), (131072, 100, 300, 200, 17)])
def test_mlp3_backward_splitk_over_many_rows(K, rows, D, H1, H2, OUT):
    """the stems' MLP backward over B x T rows: weight-gradient rows cut into chunks + ONE segmented reduce
    (smx_mlp3_backward_splitk_f32) against a float64 statement of loss.backward() and against the unsplit entry point;
    few rows (no workspace needed) take the unsplit path and must equal it bit for bit; a raised stop flag leaves the
    gradients untouched"""
    _, nd = make_net(D, H1, H2, OUT, rows + D, 'cuda')
    g = torch.Generator(device='cuda').manual_seed(rows)
    x = torch.randn(rows, D, device='cuda', generator=g)
    f = lambda *s: torch.empty(*s, device='cuda')  # noqa: E731
    h1, h2, out = f(rows, H1), f(rows, H2), f(rows, OUT)
    K.mlp3_forward(nd, x, h1, h2, out, L.SMX_ACT_NONE)
    dz3 = torch.randn(rows, OUT, device='cuda', generator=g) / rows
    n = nd.numel
    nws = K.mlp3_backward_ws_floats(nd, rows)
    assert (nws > 0) == (rows >= 2048) and nws % n == 0
    ws = f(max(nws, 1))
    g_split, g_plain = torch.full((n,), float('nan'), device='cuda'), torch.zeros(n, device='cuda')
    dz2a, dz1a, dz2b, dz1b = f(rows, H2), f(rows, H1), f(rows, H2), f(rows, H1)
    K.mlp3_backward(nd, x, h1, h2, dz3, dz2a, dz1a, g_split, None, ws=ws)
    K.mlp3_backward(nd, x, h1, h2, dz3, dz2b, dz1b, g_plain, None)
    assert torch.equal(dz2a, dz2b) and torch.equal(dz1a, dz1b)
    v = nd.views
    d3 = dz3.double()
    d2 = (d3 @ v['W3'].double()) * (h2 > 0)
    d1 = (d2 @ v['W2'].double()) * (h1 > 0)
    want = torch.cat([t.reshape(-1) for t in (d1.t() @ x.double(), d1.sum(0), d2.t() @ h1.double(), d2.sum(0),
                                              d3.t() @ h2.double(), d3.sum(0))]).float()
    close(g_split, want, atol=2e-6, rtol=2e-5, msg='split-K grads vs float64')
    close(g_plain, want, atol=2e-6, rtol=2e-5, msg='unsplit grads vs float64')
    if nws == 0:
        assert torch.equal(g_split, g_plain)
    stop = torch.ones(1, dtype=torch.int32, device='cuda')
    g_split.fill_(3.0)
    K.mlp3_backward(nd, x, h1, h2, dz3, dz2a, dz1a, g_split, None, stop, ws=ws)
    assert float(g_split.min()) == 3.0 and float(g_split.max()) == 3.0


@pytest.mark.parametrize('rows,D,H1,H2,OUT', [
    (126976, 100, 300, 200, 6), (30000, 100, 300, 200, 17), (25000, 100, 300, 200, 1), (24576 + 5, 64, 128, 96, 3),
    (25003, 128, 320, 224, 32), (40000, 36, 300, 200, 8)])
def test_mlp3_backward_fused_data_gradients_over_many_rows(K, rows, D, H1, H2, OUT):
    """K.mlp3_backward(ws=, packT=, dx=) from FUSED_ROWS_MIN rows on: dz3 -> dz2 -> dz1 -> dx as ONE launch
    (smx_mlp3_backward_rows_f32) + the register-resident weight gradients, against the layered launches (+ the separate
    dz1 . W1 product) and against a float64 statement of loss.backward(); the stop flag"""
    _, nd = make_net(D, H1, H2, OUT, rows + D, 'cuda')
    g = torch.Generator(device='cuda').manual_seed(rows)
    x = torch.randn(rows, D, device='cuda', generator=g)
    f = lambda *s: torch.empty(*s, device='cuda')  # noqa: E731
    h1, h2, out = f(rows, H1), f(rows, H2), f(rows, OUT)
    K.mlp3_forward(nd, x, h1, h2, out, L.SMX_ACT_NONE)
    dz3 = torch.randn(rows, OUT, device='cuda', generator=g) / rows
    n = nd.numel
    ws = f(K.mlp3_backward_ws_floats(nd, rows))
    npt = K.mlp3_dgrad_rows_ws_floats(nd)
    assert npt > 0 and rows >= K.FUSED_ROWS_MIN
    packT = f(npt)
    gf, gl = torch.full((n,), float('nan'), device='cuda'), torch.zeros(n, device='cuda')
    dz2f, dz1f, dxf = torch.full((rows, H2), 7.0, device='cuda'), torch.full((rows, H1), 7.0, device='cuda'), \
        torch.full((rows, D), 7.0, device='cuda')
    dz2l, dz1l, dxl = f(rows, H2), f(rows, H1), f(rows, D)
    assert K.mlp3_backward(nd, x, h1, h2, dz3, dz2f, dz1f, gf, None, ws=ws, packT=packT, dx=dxf) is True
    assert not K.mlp3_backward(nd, x, h1, h2, dz3, dz2l, dz1l, gl, None, ws=ws)
    K.linear(dz1l, 1, nd.views['W1'], 0, None, dxl, rows, D, H1)
    scale = lambda t: float(t.abs().max())  # noqa: E731
    for got, ref, what in ((dz2f, dz2l, 'dz2'), (dz1f, dz1l, 'dz1'), (dxf, dxl, 'dx'), (gf, gl, 'grads')):
        sc = scale(ref)
        close(got / sc, ref / sc, atol=2e-6, rtol=2e-5, msg=what + ' vs layered')
    v = nd.views
    d3 = dz3.double()
    d2 = (d3 @ v['W3'].double()) * (h2 > 0)
    d1 = (d2 @ v['W2'].double()) * (h1 > 0)
    dx64 = d1 @ v['W1'].double()
    want = torch.cat([t.reshape(-1) for t in (d1.t() @ x.double(), d1.sum(0), d2.t() @ h1.double(), d2.sum(0),
                                              d3.t() @ h2.double(), d3.sum(0))]).float()
    close(gf, want, atol=2e-6, rtol=2e-5, msg='grads vs float64')
    sc = scale(dx64)
    close(dxf / sc, (dx64 / sc).float(), atol=2e-6, rtol=2e-5, msg='dx vs float64')
    close(dz1f / scale(d1), (d1 / scale(d1)).float(), atol=2e-6, rtol=2e-5, msg='dz1 vs float64')
    # without dx (a stem that needs no input gradient): same dz / grads, nothing else written
    gf2 = torch.zeros(n, device='cuda')
    assert K.mlp3_backward(nd, x, h1, h2, dz3, dz2f, dz1f, gf2, None, ws=ws, packT=packT) is False
    assert torch.equal(gf2, gf)
    stop = torch.ones(1, dtype=torch.int32, device='cuda')
    gf.fill_(3.0); dxf.fill_(4.0)
    K.mlp3_backward(nd, x, h1, h2, dz3, dz2f, dz1f, gf, None, stop, ws=ws, packT=packT, dx=dxf)
    assert float(gf.min()) == 3.0 and float(gf.max()) == 3.0 and float(dxf.min()) == 4.0 and float(dxf.max()) == 4.0


@pytest.mark.parametrize('G,T0,T1,D,H1,H2,OUT,z', [
    (8, 12, 1, 11, 24, 16, 1, True), (37, 19, 1, 29, 40, 24, 1, True),
    (5, 7, 0, 16, 64, 64, 6, False), (64, 128, 1, 17, 300, 200, 1, True),
    (16, 9, 1, 376, 300, 200, 1, True), (9, 25, 0, 100, 300, 200, 17, False),
    (1, 1, 1, 3, 5, 4, 2, True), (300, 3, 1, 376, 300, 200, 1, False)])
def test_mlp3_forward_fused(K, G, T0, T1, D, H1, H2, OUT, z):
    nc, nd = make_net(D, H1, H2, OUT, G + D, 'cuda')
    g = torch.Generator().manual_seed(G * 7 + T0)
    xm = torch.randn(G, T0, D, generator=g) * 2 + 0.3
    xt = torch.randn(G, T1, D, generator=g) if T1 else None
    zm = (torch.randn(D, generator=g) * 0.3) if z else None
    zs = (torch.rand(D, generator=g) + 0.5) if z else None
    act = L.SMX_ACT_NONE if OUT == 1 else L.SMX_ACT_TANH
    rows = G * (T0 + T1)
    pc = torch.empty(C.mlp3_packed_numel(nc))
    C.mlp3_pack(nc, pc)
    oc = torch.empty(rows * OUT)
    C.mlp3_forward_fused(pc, nc, xm, xt, zm, zs, oc, act)
    pd = torch.empty(K.mlp3_packed_numel(nd)).cuda()
    K.mlp3_pack(nd, pd)
    od = torch.full((rows * OUT,), float('nan')).cuda()
    K.mlp3_forward_fused(pd, nd, dev(xm), dev(xt) if T1 else None, dev(zm) if z else None,
                         dev(zs) if z else None, od, act)
    close(od, oc, msg='fused mlp G=%d T0=%d T1=%d D=%d' % (G, T0, T1, D))


@pytest.mark.parametrize('seed', range(6))
def test_mlp3_forward_fused_random_shapes_agree_with_the_32_row_kernel(K, seed):
    """the 16-row kernel (smx_mlp3_rows16.hip) over random shapes inside its fast path -- hidden sizes that end in a
    partial 16-feature tile, D = 4 .. 400 in steps of 4, 1 .. 16 outputs, ragged row counts, with and without the
    obs_next tail and the z-filter -- against the CPU double"""
    g = torch.Generator().manual_seed(100 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))       # noqa: E731
    D, H1, H2 = 4 * ri(1, 100), ri(65, 320), ri(65, 224)
    OUT = 1 if seed % 2 == 0 else ri(2, 16)
    G, T0, T1 = ri(1, 40), ri(1, 23), ri(0, 1)
    z = bool(seed % 3)
    nc, nd = make_net(D, H1, H2, OUT, 7 + seed, 'cuda')
    xm = torch.randn(G, T0, D, generator=g) * 2 + 0.3
    xt = torch.randn(G, T1, D, generator=g) if T1 else None
    zm = (torch.randn(D, generator=g) * 0.3) if z else None
    zs = (torch.rand(D, generator=g) + 0.5) if z else None
    act = L.SMX_ACT_NONE if OUT == 1 else L.SMX_ACT_TANH
    rows = G * (T0 + T1)
    pc = torch.empty(C.mlp3_packed_numel(nc))
    C.mlp3_pack(nc, pc)
    oc = torch.empty(rows * OUT)
    C.mlp3_forward_fused(pc, nc, xm, xt, zm, zs, oc, act)
    pd = torch.empty(K.mlp3_packed_numel(nd)).cuda()
    K.mlp3_pack(nd, pd)
    od = torch.full((rows * OUT,), float('nan')).cuda()
    K.mlp3_forward_fused(pd, nd, dev(xm), dev(xt) if T1 else None, dev(zm) if z else None, dev(zs) if z else None, od, act)
    close(od, oc, msg='fused mlp D=%d H1=%d H2=%d OUT=%d rows=%d z=%s' % (D, H1, H2, OUT, rows, z))


def test_fused_exact_zfilter_switch(K):
    """the two z-filter arithmetics of the fused critic pass: the default multiplies by a
    reciprocal, the switch selects the reference's division; both within the contract, and within a few ulp of the
    filtered input of each other"""
    G, T0, D, H1, H2 = 16, 9, 376, 300, 200
    nc, nd = make_net(D, H1, H2, 1, 11, 'cuda')
    g = torch.Generator().manual_seed(2)
    xm = torch.randn(G, T0, D, generator=g) * 2 + 0.3
    zm, zs = torch.randn(D, generator=g) * 0.3, torch.rand(D, generator=g) + 0.5
    pc = torch.empty(C.mlp3_packed_numel(nc))
    C.mlp3_pack(nc, pc)
    oc = torch.empty(G * T0)
    C.mlp3_forward_fused(pc, nc, xm, None, zm, zs, oc, L.SMX_ACT_NONE)
    pd = torch.empty(K.mlp3_packed_numel(nd)).cuda()
    K.mlp3_pack(nd, pd)
    outs = []
    try:
        for exact in (False, True):
            K.fused_exact_zfilter(exact)
            od = torch.full((G * T0,), float('nan')).cuda()
            K.mlp3_forward_fused(pd, nd, dev(xm), None, dev(zm), dev(zs), od, L.SMX_ACT_NONE)
            close(od, oc, msg='fused, exact z-filter %s' % exact)
            outs.append(od.cpu())
    finally:
        K.fused_exact_zfilter(False)
    assert float((outs[0] - outs[1]).abs().max()) < 2e-6


def test_mlp3_forward_fused_full_size_property(K):
    """BASELINE full size (1024 x 129 rows x 376): the fused kernel must agree with the layered
    kernels (independent code path) on every row -- no CPU oracle needed at this size."""
    B, N, D = 1024, 128, 376
    nc, nd = make_net(D, 300, 200, 1, 99, 'cuda')
    g = torch.Generator(device='cuda').manual_seed(5)
    obs = torch.randn(B, N, D, generator=g, device='cuda')
    obs_next = torch.randn(B, 1, D, generator=g, device='cuda')
    pd = torch.empty(K.mlp3_packed_numel(nd)).cuda()
    K.mlp3_pack(nd, pd)
    out = torch.empty(B * (N + 1)).cuda()
    K.mlp3_forward_fused(pd, nd, obs, obs_next, None, None, out, L.SMX_ACT_NONE)
    x = torch.cat([obs, obs_next], 1).reshape(-1, D).contiguous()
    h1, h2 = torch.empty(x.shape[0], 300).cuda(), torch.empty(x.shape[0], 200).cuda()
    ref = torch.empty(x.shape[0], 1).cuda()
    K.mlp3_forward(nd, x, h1, h2, ref, L.SMX_ACT_NONE)
    close(out, ref.view(-1), msg='fused vs layered at full size')


@pytest.mark.parametrize('mode', [L.SMX_PPO_CLIP, L.SMX_PPO_ADAPT])
@pytest.mark.parametrize('rows,A,on_policy', [(8, 3, True), (37, 5, False), (1024, 17, True), (130, 17, False)])
def test_policy_loss_and_finalize(K, mode, rows, A, on_policy):
    g = torch.Generator().manual_seed(rows + A)
    N = 4
    log_var = torch.full((A,), -1.0) + 0.1 * torch.randn(A, generator=g)
    std = torch.exp(log_var)
    mean = torch.tanh(0.1 * torch.randn(rows, A, generator=g))
    if on_policy:
        mb = mean + 0.05 * torch.randn(rows, A, generator=g)
        act0 = mb + std * torch.randn(rows, A, generator=g)
    else:
        mb = torch.tanh(torch.randn(rows, A, generator=g))
        act0 = torch.randn(rows, A, generator=g).clamp(-1, 1)
    actions = torch.randn(rows, N, A, generator=g)
    actions[:, 0] = act0
    pds = torch.rand(rows, N, 2 * A, generator=g) + 0.2
    pds[:, 0, :A], pds[:, 0, A:] = mb, std * 1.1
    ref = torch.cat([mean + 0.02 * torch.randn(rows, A, generator=g), (std * 0.9).expand(rows, A)], 1)
    adv = torch.randn(rows, generator=g)
    ctrl = torch.zeros(L.CTRL_WORDS)
    ctrl[L.C_BETA], ctrl[L.C_ETA], ctrl[L.C_CLIP_EPS], ctrl[L.C_KL_TARGET] = 1.0, 250.0, 0.2, 1e-3
    ctrl_d = ctrl.clone().cuda()
    nblk, stride = C.loss_blocks(rows), 8 + 2 * A
    assert K.loss_blocks(rows) == nblk
    outs = {}
    for name, KK, to in (('cpu', C, lambda t: t), ('hip', K, dev)):
        gs, gk = to(torch.empty(rows, A)), to(torch.empty(rows, A))
        part = to(torch.zeros(nblk, stride))
        c = ctrl if name == 'cpu' else ctrl_d
        KK.policy_loss(mode, to(mean), to(log_var), to(actions)[:, 0, :], to(pds)[:, 0, :], to(ref),
                       to(adv), c, gs, gk, part)
        dz3, dlv, dq = to(torch.empty(rows, A)), to(torch.empty(A)), to(torch.empty(1))
        st = to(torch.zeros(L.PS_STRIDE))
        KK.policy_finalize(mode, part, nblk, gs, gk, to(log_var), rows, c, True, True, dz3, dlv, dq, st)
        outs[name] = (gs, gk, part, dz3, dlv, dq, st, c)
    for i, nm in enumerate(('g_surr', 'g_kl', 'partials', 'dz3', 'dlogvar', 'dlogvar_sumsq', 'stats')):
        a, b = outs['hip'][i], outs['cpu'][i]
        tol = 2e-4 if nm == 'partials' else ATOL     # partial SUMS over 64 rows
        close(a, b, atol=tol, rtol=1e-4 if nm in ('partials', 'dlogvar_sumsq') else RTOL, msg=nm)
    ci_d, ci_c = outs['hip'][7].cpu().view(torch.int32), outs['cpu'][7].view(torch.int32)
    assert ci_d.tolist() == ci_c.tolist()           # stop flag / step counters agree


@pytest.mark.parametrize('mode', ['clip', 'adapt'])
@pytest.mark.parametrize('rows,A,kl_target', [(37, 5, 1e-3), (1024, 17, 1e-3), (130, 17, 10.0), (7936, 6, 0.015)])
def test_policy_loss_and_finalize_against_the_reference_restatement(K, mode, rows, A, kl_target):
    """the layered loss launches (the stem policies' and the data-parallel path's: smx_ppo_policy_loss_f32 +
    smx_ppo_loss_finalize_f32) against oracle/ppo_oracle.py's _clip_loss / _adapt_loss (surreal/learner/ppo.py:194-285)
    in float64 with autograd through tanh: d loss / d z3 for every row, d loss / d log_var, and the statistics -- with the
    KL cutoff term active (kl_target 1e-3 / 0.015) and not (10)"""
    import ppo_oracle
    g = torch.Generator().manual_seed(rows * 3 + A)
    log_var = torch.full((A,), -1.0) + 0.1 * torch.randn(A, generator=g)
    std = torch.exp(log_var)
    z3 = 0.3 * torch.randn(rows, A, generator=g)
    mean = torch.tanh(z3)
    mb = mean + 0.05 * torch.randn(rows, A, generator=g)
    actions = (mb + std * torch.randn(rows, A, generator=g)).clamp(-1, 1)
    behave = torch.cat([mb, (std * 1.1).expand(rows, A)], 1).contiguous()
    ref = torch.cat([mean + 0.02 * torch.randn(rows, A, generator=g), (std * 0.9).expand(rows, A)], 1).contiguous()
    adv = torch.randn(rows, generator=g)
    m = L.SMX_PPO_ADAPT if mode == 'adapt' else L.SMX_PPO_CLIP
    ctrl = torch.zeros(L.CTRL_WORDS)
    ctrl[L.C_BETA], ctrl[L.C_ETA], ctrl[L.C_CLIP_EPS], ctrl[L.C_KL_TARGET] = 1.0, 250.0, 0.2, kl_target
    ctrl = ctrl.cuda()
    nblk = K.loss_blocks(rows)
    gs, gk, part = torch.empty(rows, A).cuda(), torch.empty(rows, A).cuda(), torch.zeros(nblk, 8 + 2 * A).cuda()
    K.policy_loss(m, dev(mean), dev(log_var), dev(actions), dev(behave), dev(ref), dev(adv), ctrl, gs, gk, part)
    dz3, dlv, dq, st = torch.empty(rows, A).cuda(), torch.empty(A).cuda(), torch.empty(1).cuda(), torch.zeros(L.PS_STRIDE).cuda()
    K.policy_finalize(m, part, nblk, gs, gk, dev(log_var), rows, ctrl, False, True, dz3, dlv, dq, st)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        O = ppo_oracle.OraclePPOLearner.__new__(ppo_oracle.OraclePPOLearner)     # the losses only: a stand-in model
        O.pd, O.cells, O.beta, O.eta, O.kl_target, O.clip_epsilon = ppo_oracle.DiagGauss(A), None, 1.0, 250.0, kl_target, 0.2
        z = z3.double().clone().requires_grad_(True)
        lv = log_var.double().clone().view(1, A).requires_grad_(True)

        class Stand(object):
            def forward_actor(self, obs, cells=None):        # builders.py:114-132 with the MLP's output given
                mu = torch.tanh(z)
                return torch.cat((mu, torch.exp(lv) * torch.ones(mu.size())), dim=1)
        O.model = Stand()
        if mode == 'adapt':
            loss, stats = O._adapt_loss(None, actions.double(), adv.double(), behave.double(), ref.double())
        else:
            loss, stats = O._clip_loss(None, actions.double(), adv.double(), behave.double())
        loss.backward()
    finally:
        torch.set_default_dtype(prev)
    # gradients scale with the cutoff coefficient when it is active: tolerances relative to the largest entry
    for got, want, nm in ((dz3, z.grad, 'd loss / d z3'), (dlv, lv.grad.view(-1), 'd loss / d log_var')):
        tol = 2e-5 * float(want.abs().max()) + 1e-10
        err = float((got.cpu().double() - want).abs().max())
        assert err <= tol, (nm, err, tol)
    s = st.cpu()
    np.testing.assert_allclose(float(s[L.PS_SURR]), stats['_surr_loss'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(float(s[L.PS_ENTROPY]), stats['_entropy'], rtol=2e-5)
    np.testing.assert_allclose(float(s[L.PS_LOSS]), stats['_kl_loss_adapt' if mode == 'adapt' else '_clip_surr_loss'],
                               rtol=5e-5, atol=2e-6)
    if mode == 'adapt':
        np.testing.assert_allclose(float(s[L.PS_KL]), stats['_pol_kl'], rtol=2e-5, atol=1e-7)


def test_value_loss(K):
    g = torch.Generator().manual_seed(11)
    for rows in (8, 37, 1024, 1500):
        v = torch.randn(rows, generator=g) * 2
        r = torch.randn(rows, generator=g) * 8 + 3
        nblk = C.value_loss_blocks(rows)
        res = {}
        for name, KK, to in (('cpu', C, lambda t: t), ('hip', K, dev)):
            dz, part = to(torch.empty(rows)), to(torch.zeros(1, nblk, 8))
            ctrl = to(torch.zeros(L.CTRL_WORDS))
            KK.value_loss(to(v), to(r), rows, dz, part[0], ctrl, True)
            st = to(torch.zeros(1, L.VS_STRIDE))
            KK.value_finalize(part, 1, nblk, st, L.VS_STRIDE)
            res[name] = (dz, st, ctrl)
        close(res['hip'][0], res['cpu'][0]), close(res['hip'][1], res['cpu'][1], rtol=1e-5)
        # against torch's own formulas (ppo.py:325-326)
        ev = 1 - torch.var(r - v) / torch.var(r)
        np.testing.assert_allclose(float(res['hip'][1][0, L.VS_EXPVAR]), float(ev), atol=1e-5)
        np.testing.assert_allclose(float(res['hip'][1][0, L.VS_LOSS]), float(((v - r) ** 2).mean()), rtol=1e-5)
        assert int(res['hip'][2].cpu().view(torch.int32)[L.C_STEP_CRITIC]) == 1


def test_uniform_gather_multi_is_uniform_indices_plus_gather_rows(K):
    """UniformReplay.sample in one launch (smx_uniform_gather_multi): the rows it draws are the ones smx_uniform_indices
    draws with the same (length, seed, offset) -- same Philox counters -- and every field (fp32 rows of odd and wide
    widths, uint8 camera frames, one-float rows) equals smx_gather_rows over those indices; injected indices (incl. out
    of range: clamped like smx_gather_rows) take precedence"""
    cap, rows = 5000, 777
    g = torch.Generator(device='cuda').manual_seed(3)
    tabs = [torch.randn(cap, 17, device='cuda', generator=g), torch.randn(cap, 4100, device='cuda', generator=g),
            torch.randn(cap, 1, device='cuda', generator=g), torch.randn(cap, 6, device='cuda', generator=g),
            torch.randint(0, 256, (cap, 3 * 20 * 20), device='cuda', dtype=torch.uint8, generator=g)]
    for length, seed, offset in ((cap, 12345, 0), (1234, 2 ** 40 + 7, 10 ** 9 + 3)):
        outs = [torch.full((rows, t.shape[1]), 0, device='cuda', dtype=t.dtype) for t in tabs]
        got_idx = torch.full((rows,), -1, dtype=torch.int64, device='cuda')
        K.uniform_gather_multi(tabs, outs, length, seed, offset, idx_out=got_idx)
        idx = torch.empty(rows, dtype=torch.int64, device='cuda')
        K.uniform_indices(idx, length, seed, offset)
        assert torch.equal(got_idx, idx) and int(idx.max()) < length and int(idx.min()) >= 0
        for t, o in zip(tabs, outs):
            ref = torch.empty_like(o)
            K.gather_rows(t, idx, ref)
            assert torch.equal(o, ref) and torch.equal(o, t[idx])
    inj = torch.randint(-3, cap + 3, (rows,), device='cuda', generator=g)
    outs = [torch.empty(rows, t.shape[1], device='cuda', dtype=t.dtype) for t in tabs]
    K.uniform_gather_multi(tabs, outs, cap, 1, 0, idx=inj)
    for t, o in zip(tabs, outs):
        assert torch.equal(o, t[inj.clamp(0, cap - 1)])


def test_clip_adam_matches_torch_optim(K):
    """three consecutive steps against torch.optim.Adam + clip_grad_norm_ on CPU"""
    g = torch.Generator().manual_seed(21)
    n = 5000
    theta0 = torch.randn(n, generator=g)
    p = torch.nn.Parameter(theta0.clone())
    opt = torch.optim.Adam([p], lr=1e-3, weight_decay=0.0)
    th, m, v = dev(theta0.clone()), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    ctrl = torch.zeros(L.CTRL_WORDS)
    ctrl[L.C_LR_ACTOR], ctrl[L.C_ACTOR_MAX_NORM] = 1e-3, 5.0
    ctrl = ctrl.cuda()
    ci = ctrl.view(torch.int32)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (0.2 if step == 2 else 0.01)   # step 2 gets clipped
        p.grad = grad.clone()
        tn = torch.nn.utils.clip_grad_norm_([p], 5.0)
        opt.step()
        ci[L.C_STEP_ACTOR] = step
        gd = dev(grad)
        nb = K.sumsq_blocks(n)
        part = torch.zeros(nb).cuda()
        K.sumsq_partials(gd, part)
        gn = torch.zeros(1).cuda()
        K.clip_adam(th, gd, m, v, part, nb, ctrl, 0, True, gn)
        np.testing.assert_allclose(float(gn), float(tn), rtol=1e-5)
        close(th, p.detach(), atol=1e-6, rtol=1e-6, msg='theta after step %d' % step)
    ci[L.C_STOP] = 1
    before = th.clone()
    K.clip_adam(th, gd, m, v, part, nb, ctrl, 0, True, gn)
    assert torch.equal(before, th)


@pytest.mark.parametrize('dtype', [torch.uint8, torch.int16])
def test_replay_kernels_byte_rows(K, dtype):
    """ring insert / gather / window emission of non-fp32 rows (uint8 camera frames): bit-exact against index
    arithmetic, for row pitches that move as 16-byte lanes (3 x 84 x 84 B), dwords and single bytes, rows longer
    than one 8 KB segment, and unaligned sub-views"""
    g = torch.Generator().manual_seed(7)
    hi = 255 if dtype == torch.uint8 else 30000
    for cap, width in ((5, 7), (9, 12), (40, 3 * 84 * 84), (33, 8192 * 2 + 16), (17, 1001)):
        table_c = torch.zeros(cap, width, dtype=dtype)
        table_d = table_c.clone().cuda()
        cursor = 0
        for n in (3, cap // 2 + 1, 2):
            src = torch.randint(0, hi, (n, width), generator=g).to(dtype)
            C.ring_insert(table_c, cursor, src)
            K.ring_insert(table_d, cursor, dev(src))
            cursor = (cursor + n) % cap
        assert torch.equal(table_d.cpu(), table_c)
        idx = torch.randint(0, cap, (19,), generator=g)
        dc, dd = torch.empty(19, width, dtype=dtype), torch.empty(19, width, dtype=dtype).cuda()
        C.gather_rows(table_c, idx, dc)
        K.gather_rows(table_d, dev(idx), dd)
        assert torch.equal(dd.cpu(), dc)
    for width in (3 * 8 * 8, 7):
        src = torch.randint(0, hi, (4, 14, width), generator=g).to(dtype)
        W = (14 - 5) // 3 + 1
        dc, dd = torch.empty(4 * W, 5, width, dtype=dtype), torch.empty(4 * W, 5, width, dtype=dtype).cuda()
        C.window_emit(src, 0, 5, 3, W, dc)
        K.window_emit(dev(src), 0, 5, 3, W, dd)
        assert torch.equal(dd.cpu(), dc)


def test_uint8_frames_stay_uint8_in_the_device_replay(K):
    """pixel observations in the replay's device tier: uint8 tables (a quarter of the fp32 bytes), FIFO pop and
    uniform gather return the frames bit-exact and still uint8 -- what the CNN stem's im2col reads"""
    from surreal_amd.replay import FIFOReplay, UniformReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    lc, ec, sc = ppo_learner_config(), ppo_env_config(6, 2), ppo_session_config('/tmp/surreal_amd_test')
    lc.replay.memory_size, lc.replay.batch_size = 13, 4
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 255, (10, 4, 3, 12, 12), generator=g).to(torch.uint8).cuda()
    low = torch.randn(10, 4, 6, generator=g).cuda()
    f = FIFOReplay(lc, ec, sc)
    f.insert_batch({'frames': frames, 'low': low})
    assert f._tables['frames'].data.dtype == torch.uint8 and f._tables['low'].data.dtype == torch.float32
    assert f._tables['frames'].data.numel() * 1 == (13 + 3) * 4 * 3 * 12 * 12
    b = f.sample_batch(4)
    assert b['frames'].dtype == torch.uint8 and torch.equal(b['frames'], frames[:4]) and torch.equal(b['low'], low[:4])
    f.insert_batch({'frames': frames[:9], 'low': low[:9]})         # wraps the ring
    b = f.sample_batch(8)
    assert torch.equal(b['frames'], torch.cat([frames[4:], frames[:2]]))
    u = UniformReplay(lc, ec, sc)
    u.insert_batch({'frames': frames, 'low': low})
    got = u.sample_batch(5, indices=[9, 0, 3, 3, 7])
    assert got['frames'].dtype == torch.uint8 and torch.equal(got['frames'], frames[[9, 0, 3, 3, 7]])


@pytest.mark.parametrize('F,C,H,W,k,st,cout', [(5, 3, 84, 84, 8, 4, 16), (3, 2, 20, 24, 8, 4, 16), (7, 1, 28, 36, 8, 4, 9),
                                                (2, 4, 84, 84, 8, 4, 16), (1, 4, 12, 12, 4, 4, 5)])
def test_conv_u8_forward_is_im2col_plus_gemm(K, F, C, H, W, k, st, cout):
    """the implicit-GEMM first convolution over uint8 frames (smx_conv_u8_forward_f32) against the materialised
    route (smx_im2col_f32 with /255 + smx_linear_f32 with ReLU) and the CPU double; every byte value occurs, a
    row count that is not a multiple of 16, fewer than 16 output channels, and the stop flag"""
    g = torch.Generator().manual_seed(F * 7 + C)
    frames = torch.randint(0, 256, (F, C, H, W), generator=g).to(torch.uint8)
    frames.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
    Kc = C * k * k
    Wt = torch.randn(cout, C, k, k, generator=g) / Kc ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    Ho, Wo = (H - k) // st + 1, (W - k) // st + 1
    rows = F * Ho * Wo
    assert K.conv_u8_supported(dev(frames), C, H, W, k, st, cout)
    yc = torch.empty(rows, cout)
    C_ = C
    from cpu_kernels import TorchCpuKernels
    TorchCpuKernels().conv_u8_forward(frames, F, C_, H, W, k, st, Wt, b, cout, yc)
    yd = torch.full((rows, cout), float('nan')).cuda()
    K.conv_u8_forward(dev(frames), F, C, H, W, k, st, dev(Wt), dev(b), cout, yd)
    close(yd, yc, msg='implicit conv vs double')
    cols = torch.empty(rows, Kc).cuda()
    K.im2col(dev(frames), F, C, H, W, k, st, cols, scale_div=255.0)
    ym = torch.empty(rows, cout).cuda()
    K.linear(cols, 1, dev(Wt).view(cout, Kc), 1, dev(b), ym, rows, cout, Kc, act=L.SMX_ACT_RELU)
    close(yd, ym.cpu(), atol=2e-6, rtol=2e-6, msg='implicit conv vs im2col + GEMM')
    stop = torch.ones(1, dtype=torch.int32).cuda()
    yd.fill_(3.0)
    K.conv_u8_forward(dev(frames), F, C, H, W, k, st, dev(Wt), dev(b), cout, yd, stop=stop)
    assert float(yd.min()) == 3.0


@pytest.mark.parametrize('F,C,H,W,k,st,cout', [(5, 3, 84, 84, 8, 4, 16), (3, 2, 20, 24, 8, 4, 16), (7, 1, 28, 36, 8, 4, 9),
                                                (40, 3, 84, 84, 8, 4, 16), (1, 4, 12, 12, 4, 4, 5)])
def test_conv_u8_wgrad_is_im2col_plus_wgrad_gemm(K, F, C, H, W, k, st, cout):
    """the implicit first-convolution weight gradient (smx_conv_u8_wgrad_f32) against the CPU double (patch matrix,
    dy^T . patches in fp32) and against the materialised HIP route; fp32 tolerance on sums of F*Ho*Wo terms"""
    g = torch.Generator().manual_seed(F * 3 + C)
    frames = torch.randint(0, 256, (F, C, H, W), generator=g).to(torch.uint8)
    Kc = C * k * k
    Ho, Wo = (H - k) // st + 1, (W - k) // st + 1
    rows = F * Ho * Wo
    dy = torch.randn(rows, cout, generator=g) / rows ** 0.5
    dy[torch.rand(rows, cout, generator=g) < 0.4] = 0.0          # a ReLU mask's zeros
    from cpu_kernels import TorchCpuKernels
    Wc, bc = torch.empty(cout, Kc), torch.empty(cout)
    TorchCpuKernels().conv_u8_wgrad(frames, F, C, H, W, k, st, dy, cout, Wc, bc, None)
    ws = torch.empty(K.conv_u8_wgrad_ws_floats(cout, Kc)).cuda()
    Wd, bd = torch.full((cout, Kc), float('nan')).cuda(), torch.full((cout,), float('nan')).cuda()
    K.conv_u8_wgrad(dev(frames), F, C, H, W, k, st, dev(dy), cout, Wd, bd, ws)
    close(Wd, Wc, atol=2e-5, rtol=2e-5, msg='implicit conv wgrad vs double')
    close(bd, bc, atol=2e-5, rtol=2e-5, msg='bias gradient')
    cols = torch.empty(rows, Kc).cuda()
    K.im2col(dev(frames), F, C, H, W, k, st, cols, scale_div=255.0)
    Wm, bm = torch.empty(cout, Kc).cuda(), torch.empty(cout).cuda()
    K.linear_wgrad(dev(dy), cols, Wm, bm, cout, Kc, rows)
    close(Wd, Wm.cpu(), atol=2e-5, rtol=2e-5, msg='implicit vs materialised wgrad')
    stop = torch.ones(1, dtype=torch.int32).cuda()
    Wd.fill_(3.0)
    K.conv_u8_wgrad(dev(frames), F, C, H, W, k, st, dev(dy), cout, Wd, bd, ws, stop=stop)
    assert float(Wd.min()) == 3.0


@pytest.mark.parametrize('F,H,W,k,st,cout', [(5, 20, 20, 4, 2, 32), (3, 9, 11, 3, 1, 20), (40, 20, 20, 4, 2, 32),
                                              (2, 6, 5, 2, 1, 7), (1, 20, 20, 4, 2, 16)])
def test_conv_cl_forward_and_wgrad_are_im2col_plus_gemm(K, F, H, W, k, st, cout):
    """the implicit second convolution (fp32 channel-last source of 16 channels): forward + bias + ReLU and the weight /
    bias gradients against the CPU double (materialised patches) and the materialised HIP route"""
    C = 16
    g = torch.Generator().manual_seed(F * 5 + k)
    src = torch.relu(torch.randn(F, H * W, C, generator=g))
    Kc = C * k * k
    Wt = torch.randn(cout, C, k, k, generator=g) / Kc ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    Ho, Wo = (H - k) // st + 1, (W - k) // st + 1
    rows = F * Ho * Wo
    from cpu_kernels import TorchCpuKernels
    Cd = TorchCpuKernels()
    assert K.conv_cl_supported(dev(src), C, k, cout)
    yc = torch.empty(rows, cout)
    Cd.conv_cl_forward(src, F, C, H, W, k, st, Wt, b, cout, yc)
    yd = torch.full((rows, cout), float('nan')).cuda()
    K.conv_cl_forward(dev(src), F, C, H, W, k, st, dev(Wt), dev(b), cout, yd)
    close(yd, yc, msg='implicit conv2 forward vs double')
    cols = torch.empty(rows, Kc).cuda()
    K.im2col(dev(src), F, C, H, W, k, st, cols, channel_last=True)
    ym = torch.empty(rows, cout).cuda()
    K.linear(cols, 1, dev(Wt).view(cout, Kc), 1, dev(b), ym, rows, cout, Kc, act=L.SMX_ACT_RELU)
    close(yd, ym.cpu(), atol=2e-6, rtol=2e-6, msg='implicit vs im2col + GEMM')
    dy = torch.randn(rows, cout, generator=g) / rows ** 0.5
    dy[torch.rand(rows, cout, generator=g) < 0.4] = 0.0
    Wc, bc = torch.empty(cout, Kc), torch.empty(cout)
    Cd.conv_cl_wgrad(src, F, C, H, W, k, st, dy, cout, Wc, bc, None)
    ws = torch.empty(K.conv_cl_wgrad_ws_floats(cout, k)).cuda()
    Wd, bd = torch.full((cout, Kc), float('nan')).cuda(), torch.full((cout,), float('nan')).cuda()
    K.conv_cl_wgrad(dev(src), F, C, H, W, k, st, dev(dy), cout, Wd, bd, ws)
    close(Wd, Wc, atol=2e-5, rtol=2e-5, msg='implicit conv2 wgrad vs double')
    close(bd, bc, atol=2e-5, rtol=2e-5, msg='bias gradient')
    stop = torch.ones(1, dtype=torch.int32).cuda()
    yd.fill_(3.0); Wd.fill_(3.0)
    K.conv_cl_forward(dev(src), F, C, H, W, k, st, dev(Wt), dev(b), cout, yd, stop=stop)
    K.conv_cl_wgrad(dev(src), F, C, H, W, k, st, dev(dy), cout, Wd, bd, ws, stop=stop)
    assert float(yd.min()) == 3.0 and float(Wd.min()) == 3.0


@pytest.mark.parametrize('F,H,W,k,st,cout', [(5, 20, 20, 4, 2, 32), (3, 9, 11, 2, 1, 20), (40, 20, 20, 4, 2, 32),
                                              (2, 7, 6, 2, 1, 7), (1, 21, 20, 4, 2, 16), (3, 13, 9, 6, 3, 12)])
def test_conv_cl_dgrad_is_gemm_plus_col2im(K, F, H, W, k, st, cout):
    """the implicit data gradient of the second convolution against dy . W + the col2im gather (CPU double and the
    materialised HIP route): odd map sizes, pixels no output position sees, fewer than 16 / 32 output channels, the
    ReLU mask, the stop flag"""
    C = 16
    g = torch.Generator().manual_seed(F * 11 + k)
    Wt = torch.randn(cout, C, k, k, generator=g) / (C * k * k) ** 0.5
    Ho, Wo = (H - k) // st + 1, (W - k) // st + 1
    rows = F * Ho * Wo
    dy = torch.randn(rows, cout, generator=g)
    act = torch.randn(F * H * W, C, generator=g)
    from cpu_kernels import TorchCpuKernels
    Cd = TorchCpuKernels()
    assert K.conv_cl_dgrad_supported(dev(dy), C, k, st, cout)
    dxc = torch.empty(F * H * W, C)
    Cd.conv_cl_dgrad(dy, F, C, H, W, k, st, Wt, cout, act, dxc)
    dxd = torch.full((F * H * W, C), float('nan')).cuda()
    K.conv_cl_dgrad(dev(dy), F, C, H, W, k, st, dev(Wt), cout, dev(act), dxd)
    close(dxd, dxc, atol=2e-5, rtol=2e-5, msg='implicit conv2 dgrad vs double')
    dcols = torch.empty(rows, C * k * k).cuda()
    K.linear(dev(dy), 1, dev(Wt).view(cout, -1), 0, None, dcols, rows, C * k * k, cout)
    dxm = torch.empty(F * H * W, C).cuda()
    K.col2im(dcols, F, C, H, W, k, st, dev(act), dxm)
    close(dxd, dxm.cpu(), atol=2e-5, rtol=2e-5, msg='implicit vs GEMM + col2im')
    stop = torch.ones(1, dtype=torch.int32).cuda()
    dxd.fill_(3.0)
    K.conv_cl_dgrad(dev(dy), F, C, H, W, k, st, dev(Wt), cout, dev(act), dxd, stop=stop)
    assert float(dxd.min()) == 3.0


def test_replay_kernels(K):
    g = torch.Generator().manual_seed(31)
    for cap, width in ((5, 7), (96, 44), (1000, 376)):
        table_c = torch.zeros(cap, width)
        table_d = table_c.clone().cuda()
        cursor = 0
        for n in (3, cap // 2 + 1, 2):
            src = torch.randn(n, width, generator=g)
            C.ring_insert(table_c, cursor, src)
            K.ring_insert(table_d, cursor, dev(src))
            cursor = (cursor + n) % cap
        assert torch.equal(table_d.cpu(), table_c)           # bit exact: byte copies
        idx = torch.randint(0, cap, (64,), generator=g)
        dc, dd = torch.empty(64, width), torch.empty(64, width).cuda()
        C.gather_rows(table_c, idx, dc)
        K.gather_rows(table_d, dev(idx), dd)
        assert torch.equal(dd.cpu(), dc)
    # uniform indices: in range, reproducible, roughly uniform (distributional parity with
    # random.randint -- the Python Mersenne stream itself is not reproducible on a GPU)
    idx = torch.empty(1 << 16, dtype=torch.int64).cuda()
    K.uniform_indices(idx, 1000, 1234, 0)
    a = idx.cpu()
    assert int(a.min()) >= 0 and int(a.max()) < 1000
    K.uniform_indices(idx, 1000, 1234, 0)
    assert torch.equal(idx.cpu(), a)
    counts = torch.bincount(a, minlength=1000).float()
    assert float(counts.std()) < 3 * np.sqrt(65.5)
    K.uniform_indices(idx, 1000, 1234, 1 << 16)
    assert not torch.equal(idx.cpu(), a)
    # ... and bit for bit the published generator: Random123's known-answer vectors through the device rounds, then
    # the sampler's indices against the numpy statement of the same construction (tests/philox_ref.py)
    import philox_ref as P
    ck = torch.tensor([list(c) + list(k) for c, k, _ in P.KAT], dtype=torch.int64)
    out = torch.zeros(len(P.KAT), 4, dtype=torch.int32).cuda()
    K.philox4x32_10((ck - ((ck >> 31) << 32)).to(torch.int32).cuda(), out)       # (uint32 bit patterns as int32)
    got = (out.cpu().to(torch.int64) & 0xFFFFFFFF).tolist()
    assert got == [list(o) for _, _, o in P.KAT], [[hex(x) for x in r] for r in got]
    for length, seed, offset in ((1000, 1234, 0), (1000000, 0x1234567890ABCDEF, (1 << 40) + 7), (3, 0, 0xFFFFFFFF)):
        small = torch.empty(512, dtype=torch.int64).cuda()
        K.uniform_indices(small, length, seed, offset)
        assert small.cpu().tolist() == P.uniform_indices(512, length, seed, offset).tolist(), (length, seed, offset)
    # window emission: n_step/stride moving window (exp_sender_wrapper.py:209-228)
    src = torch.randn(6, 14, 10, generator=g)
    W = (14 - 5) // 3 + 1
    dc, dd = torch.empty(6 * W, 5, 10), torch.empty(6 * W, 5, 10).cuda()
    C.window_emit(src, 0, 5, 3, W, dc)
    K.window_emit(dev(src), 0, 5, 3, W, dd)
    assert torch.equal(dd.cpu(), dc) and W == 4
    # obs_next of every window: start = n_step, one row
    nc, nd = torch.empty(6 * 3, 1, 10), torch.empty(6 * 3, 1, 10).cuda()
    C.window_emit(src, 5, 1, 3, 3, nc)
    K.window_emit(dev(src), 5, 1, 3, 3, nd)
    assert torch.equal(nd.cpu(), nc) and torch.equal(nc[1, 0], src[0, 8])
    # synthetic env step: bit-exact against the CPU statement, incl. episode reset
    n, D, A, T = 9, 23, 5, 4
    st = torch.randn(n, D, generator=g)
    init = torch.randn(n, D, generator=g)
    rolls_c = [torch.zeros(n, T, D), torch.zeros(n, T, A), torch.zeros(n, T), torch.zeros(n, T)]
    rolls_d = [r.clone().cuda() for r in rolls_c]
    sc, sd = st.clone(), st.clone().cuda()
    for t in range(T):
        act = torch.randn(n, A, generator=g) * 0.8
        C.synth_env_step(sc, init, act, t, 3, t, *rolls_c)
        K.synth_env_step(sd, dev(init), dev(act), t, 3, t, *rolls_d)
    assert torch.equal(sd.cpu(), sc)
    for rc, rd in zip(rolls_c, rolls_d):
        assert torch.equal(rd.cpu(), rc)


@pytest.mark.parametrize('rows,D', [(1024, 376), (96, 20), (4100, 376)])
def test_mlp3_multi_jobs_with_transposed_operands(K, rows, D):
    """actor + critic jobs in shared launches; the weight-gradient GEMMs read the transposed
    (K-contiguous) copies written by the producing epilogues; a raised stop flag masks one job.  4100 rows: the
    layers and data gradients take the LDS-tiled kernel (its transposed-copy epilogue, its stop flag, ragged rows)"""
    g = torch.Generator().manual_seed(rows)
    specs = [(D, 300, 200, 17, L.SMX_ACT_TANH), (D, 300, 200, 1, L.SMX_ACT_NONE)] if D == 376 else \
        [(D, 40, 24, 5, L.SMX_ACT_TANH), (D, 40, 24, 1, L.SMX_ACT_NONE)]
    x = torch.randn(rows, D, generator=g)

    def mk(to, stop_val):
        jobs = []
        for k, (d, h1, h2, o, act) in enumerate(specs):
            nc, nd = make_net(d, h1, h2, o, 77 + k, 'cuda')
            net = nc if to is None else nd
            t = (lambda z: z) if to is None else dev
            f = lambda *s: t(torch.zeros(*s))  # noqa: E731
            jobs.append(dict(net=net, x=t(x.clone()), h1=f(rows, h1), h2=f(rows, h2), out=f(rows, o), act=act,
                             dz3=t(torch.randn(rows, o, generator=torch.Generator().manual_seed(5 + k)) / rows),
                             dz2=f(rows, h2), dz1=f(rows, h1), grads=f(net.numel),
                             sumsq=f(K.mlp3_backward_partials(nd)), xT=t(x.t().contiguous()),
                             h1T=f(h1, rows), h2T=f(h2, rows), dz2T=f(h2, rows), dz1T=f(h1, rows)))
            jobs[-1]['dz3T'] = t(jobs[-1]['dz3'].cpu().t().contiguous())
        jobs[0]['stop'] = t(torch.tensor([stop_val], dtype=torch.int32))
        return jobs
    for stop_val in (0, 1):
        jc, jd = mk(None, stop_val), mk('cuda', stop_val)
        C.mlp3_forward_multi(jc)
        K.mlp3_forward_multi(jd)
        C.mlp3_backward_multi(jc)
        K.mlp3_backward_multi(jd)
        for a, b in zip(jd, jc):
            for key in ('h1', 'h2', 'out', 'h1T', 'h2T', 'dz2', 'dz1', 'dz2T', 'dz1T'):
                close(a[key], b[key], msg='%s stop=%d' % (key, stop_val))
            # (two fp32 sums over `rows` terms in different orders: the absolute bound grows with the row count)
            close(a['grads'], b['grads'], atol=2e-6 if rows <= 1024 else 1e-5, rtol=2e-5, msg='grads stop=%d' % stop_val)
            np.testing.assert_allclose(float(a['sumsq'].sum()), float(b['sumsq'].sum()), rtol=1e-5)
        if stop_val:
            assert float(jd[0]['out'].abs().max()) == 0.0 and float(jd[0]['grads'].abs().max()) == 0.0
            assert float(jd[1]['grads'].abs().max()) > 0.0


# ---- LSTM stem: smx_lstm_forward_f32 / smx_lstm_backward_f32 vs torch.nn.LSTM (ATen CPU fp32,
# the op the reference calls, ppo_net.py:146-149) and vs the double's statement of the contract
@pytest.mark.parametrize('B,T,D,H,cells', [
    (2, 21, 17, 100, True),      # cfg1 epochs: E = 25 - 5 + 1
    (2, 26, 17, 100, True),      # cfg1 critic pass: N + 1
    (5, 4, 7, 12, True),         # tiny golden
    (37, 9, 17, 100, False),     # ragged rows, zero initial state
    (64, 21, 17, 100, True),     # cfg2-sized batch
    (6, 7, 9, 108, True),        # 100 < H <= 112: the wider 4-row instantiation
    (1024, 3, 17, 100, True),    # B >= 512: four rows per workgroup on v_mfma_f32_4x4x1 (lstm_fwdm / bwdm): 256 workgroups
    (1027, 6, 17, 100, True),    # ... with a ragged last workgroup (3 of 4 rows)
    (1026, 4, 9, 108, True),     # ... the 100 < H <= 112 instantiation
    (1025, 3, 24, 100, False),   # ... with the input projection as a GEMM in front (D > 20)
    (1024, 1, 17, 100, True),    # ... a single step (1024 actors acting): every clamped prefetch index is 0
    (600, 2, 17, 100, True),     # ... two steps
    (40, 1, 17, 100, True), (40, 2, 17, 100, False),     # ... and the same on the vector kernels
    (1030, 5, 24, 112, True),    # ... both
    (515, 7, 17, 100, False),    # ... ragged (3 of 4 rows in the last workgroup), zero initial state
    (513, 5, 9, 128, True),      # 112 < H <= 128 at B >= 512: the vector kernels, two rows per workgroup
    (3, 5, 9, 128, True),        # H > 112: 16-row kernels, W_hh fragments re-read every step
    (20, 6, 11, 256, True),
])
def test_lstm_forward_backward_match_aten(K, B, T, D, H, cells):
    from surreal_amd.model.ppo_net import LstmParams
    torch.manual_seed(B * 1000 + T)
    ref = torch.nn.LSTM(D, H, 1, batch_first=True)
    x = torch.randn(B, T, D)
    h0, c0 = 0.3 * torch.randn(1, B, H), 0.3 * torch.randn(1, B, H)
    out, (hN, cN) = ref(x, (h0, c0) if cells else None)
    dout = torch.randn(B, T, H)
    (out * dout).sum().backward()
    plist = (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)
    flat = torch.cat([p.detach().reshape(-1) for p in plist]).cuda()
    assert flat.numel() == L.load().smx_lstm_param_count(D, H) == LstmParams.count(D, H)
    net = LstmParams(flat, 0, D, H)
    xd = x.cuda().reshape(B * T, D).contiguous()
    f = lambda *s: torch.empty(*s, device='cuda')  # noqa: E731
    gates, od, cs, hp, hNd, cNd = f(B * T, 4 * H), f(B, T, H), f(B * T, H), f(B * T, H), f(B, H), f(B, H)
    h0d = h0[0].cuda().contiguous() if cells else None
    c0d = c0[0].cuda().contiguous() if cells else None
    K.lstm_forward(net, xd, B, T, h0d, c0d, gates, od, cs, hp, hNd, cNd)
    close(od, out, msg='h_t')
    close(hNd, hN[0], msg='h_N')
    close(cNd, cN[0], msg='c_N')
    # the saved tensors are what the contract (test double) says
    flat_c = flat.cpu()
    net_c = LstmParams(flat_c, 0, D, H)
    g2, o2, c2, p2 = torch.empty(B * T, 4 * H), torch.empty(B, T, H), torch.empty(B * T, H), torch.empty(B * T, H)
    C.lstm_forward(net_c, x.reshape(B * T, D), B, T, h0[0] if cells else None, c0[0] if cells else None,
                   g2, o2, c2, p2)
    close(gates, g2, msg='activated gates')
    close(cs, c2, msg='c_t')
    close(hp, p2, msg='h_{t-1}')
    grads = torch.zeros(flat.numel(), device='cuda')
    K.lstm_backward(net, xd, B, T, c0d, gates, cs, hp, dout.cuda().contiguous(), gates, grads)   # dgates aliases gates
    gref = torch.cat([p.grad.reshape(-1) for p in plist])
    scale = float(gref.abs().max())
    close(grads.cpu() / scale, gref / scale, atol=2e-6, rtol=1e-5, msg='BPTT gradients')


def test_lstm_stop_flag_and_limits(K):
    from surreal_amd.model.ppo_net import LstmParams
    B, T, D, H = 3, 4, 5, 8
    flat = torch.randn(LstmParams.count(D, H), device='cuda')
    net = LstmParams(flat, 0, D, H)
    x = torch.randn(B * T, D, device='cuda')
    gates, out, cs = (torch.full((B * T, n), 7.0, device='cuda') for n in (4 * H, H, H))
    stop = torch.ones(1, dtype=torch.int32, device='cuda')
    K.lstm_forward(net, x, B, T, None, None, gates, out, cs, None, None, None, stop=stop)
    assert float(out.min()) == 7.0 and float(gates.min()) == 7.0          # skipped entirely
    bad = LstmParams(torch.randn(LstmParams.count(D, 6), device='cuda'), 0, D, 6)     # H % 4 != 0
    with pytest.raises(L.SmxError):
        K.lstm_forward(bad, x, B, T, None, None, gates, out, cs)


# ---- CNN stem: data-movement kernels vs the double, whole stem vs torch.nn (ATen conv2d) --------
@pytest.mark.parametrize('F,C,H,W,k,s,u8,cl', [
    (3, 3, 20, 20, 8, 4, True, False), (5, 3, 84, 84, 8, 4, True, False),
    (4, 16, 20, 20, 4, 2, False, True), (2, 2, 13, 17, 4, 2, False, False), (7, 5, 9, 9, 3, 1, False, True),
])
def test_im2col_col2im_bit_exact(K, F, C, H, W, k, s, u8, cl):
    g = torch.Generator().manual_seed(F * 100 + H)
    Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
    if u8:
        src = torch.randint(0, 256, (F, C, H, W), generator=g, dtype=torch.uint8)
    else:
        src = torch.randn((F, H * W, C) if cl else (F, C, H, W), generator=g)
    cols_d, cols_c = torch.empty(F * Ho * Wo, C * k * k, device='cuda'), torch.empty(F * Ho * Wo, C * k * k)
    div = 255.0 if u8 else 0.0
    K.im2col(src.cuda(), F, C, H, W, k, s, cols_d, channel_last=cl, scale_div=div)
    C_ = TorchCpuKernels()
    C_.im2col(src, F, C, H, W, k, s, cols_c, channel_last=cl, scale_div=div)
    assert torch.equal(cols_d.cpu(), cols_c)                       # pure data movement (+ x / 255)
    dcols = torch.randn(F * Ho * Wo, C * k * k, generator=g)
    act = torch.randn(F * H * W, C, generator=g)
    dx_d, dx_c = torch.empty(F * H * W, C, device='cuda'), torch.empty(F * H * W, C)
    K.col2im(dcols.cuda(), F, C, H, W, k, s, act.cuda(), dx_d)
    C_.col2im(dcols, F, C, H, W, k, s, act, dx_c)
    close(dx_d, dx_c, atol=1e-6, rtol=1e-6, msg='col2im gather')
    w = torch.randn(6, C * Ho * Wo, generator=g)
    a, b = torch.empty(6, C * Ho * Wo, device='cuda'), torch.empty(6, C * Ho * Wo, device='cuda')
    K.flatten_order(w.cuda(), 6, C, Ho * Wo, True, a)
    K.flatten_order(a, 6, C, Ho * Wo, False, b)
    assert torch.equal(b.cpu(), w)
    assert torch.equal(a.cpu(), w.view(6, C, Ho * Wo).transpose(1, 2).reshape(6, -1))


@pytest.mark.parametrize('implicit_wgrad', [True, False])
@pytest.mark.parametrize('F,C,H,W,feat', [(3, 3, 20, 20, 8), (5, 2, 36, 28, 24), (16, 3, 84, 84, 256)])
def test_cnn_stem_forward_backward_match_aten(K, F, C, H, W, feat, implicit_wgrad):
    """the whole stem against ATen's conv2d + autograd.  implicit_wgrad: with the split-K / partial workspace the two
    convolutions' weight gradients come from the implicit-GEMM kernels, without it from the materialised patch
    matrices; the forward passes and conv2's data gradient are implicit either way (uint8 frames, k = 2 stride)"""
    import torch.nn as nn
    from surreal_amd.model.cnn_stem import CnnParams, CnnStem
    torch.manual_seed(F + H)
    ref = nn.Sequential(nn.Conv2d(C, 16, 8, 4), nn.ReLU(), nn.Conv2d(16, 32, 4, 2), nn.ReLU(), nn.Flatten())
    with torch.no_grad():
        n_flat = ref(torch.zeros(1, C, H, W)).shape[1]
    fc = nn.Linear(n_flat, feat)
    frames = torch.randint(0, 256, (F, C, H, W), dtype=torch.uint8)
    y = torch.relu(fc(ref(frames.float() / 255.0)))
    dy = torch.randn(F, feat)
    (y * dy).sum().backward()
    flat = torch.zeros(CnnParams.count((C, H, W), feat), device='cuda')
    p = CnnParams(flat, 0, (C, H, W), feat)
    src = {'conv1.W': ref[0].weight, 'conv1.b': ref[0].bias, 'conv2.W': ref[2].weight,
           'conv2.b': ref[2].bias, 'fc.W': fc.weight, 'fc.b': fc.bias}
    for k, v in p.views.items():
        v.copy_(src[k].detach())
    stem = CnnStem(K)
    ws = stem.workspace(p, F, 'cuda')
    if implicit_wgrad:
        ws.sk = stem.splitk_workspace(p, F, 'cuda')
    D = 5                                       # features live at a column offset, as in the learner
    xin = torch.zeros(F, D + feat, device='cuda')
    stem.forward(p, frames.cuda(), F, ws, xin[:, D:])
    close(xin[:, D:], y, msg='stem features')
    dxin = torch.zeros_like(xin)
    dxin[:, D:] = (dy * (y.detach() > 0)).cuda()
    grads = torch.zeros_like(flat)
    stem.backward(p, F, ws, dxin[:, D:], grads)
    gp = CnnParams(grads, 0, (C, H, W), feat)
    for k, v in gp.views.items():
        scale = float(src[k].grad.abs().max())
        close(v.cpu() / scale, src[k].grad / scale, atol=5e-6, rtol=1e-5, msg='grad ' + k)


# ---- single-launch variants used by the lock-step epoch: they must equal the separate launches
@pytest.mark.parametrize('mode', [L.SMX_PPO_CLIP, L.SMX_PPO_ADAPT])
@pytest.mark.parametrize('rows,A', [(8, 3), (130, 5), (1024, 17), (4000, 17)])
def test_epoch_losses_single_launch_equals_separate_launches(K, mode, rows, A):
    g = torch.Generator().manual_seed(rows * 7 + A)
    log_var = (torch.full((A,), -1.0) + 0.1 * torch.randn(A, generator=g)).cuda()
    mean = torch.tanh(0.1 * torch.randn(rows, A, generator=g)).cuda()
    std = torch.exp(log_var).cpu()
    actions = (mean.cpu() + std * torch.randn(rows, A, generator=g)).cuda()
    behave = torch.cat([mean.cpu() + 0.05 * torch.randn(rows, A, generator=g), (std * 1.1).expand(rows, A)], 1).cuda()
    ref = torch.cat([mean.cpu() + 0.02 * torch.randn(rows, A, generator=g), (std * 0.9).expand(rows, A)], 1).cuda()
    adv = torch.randn(rows, generator=g).cuda()
    vals, rets = (torch.randn(rows, generator=g) * 2).cuda(), (torch.randn(rows, generator=g) * 5).cuda()
    nblk, nbv, stride = K.loss_blocks(rows), K.value_loss_blocks(rows), 8 + 2 * A
    ldT = rows + 16
    res = []
    for fused in (False, True):
        ctrl = torch.zeros(L.CTRL_WORDS)
        ctrl[L.C_BETA], ctrl[L.C_ETA], ctrl[L.C_CLIP_EPS], ctrl[L.C_KL_TARGET] = 1.0, 250.0, 0.2, 1e-3
        ctrl = ctrl.cuda()
        f = lambda *s: torch.zeros(*s, device='cuda')  # noqa: E731
        gs, gk, part, dz3, dlv, dq, st = f(rows, A), f(rows, A), f(nblk, stride), f(rows, A), f(A), f(1), f(L.PS_STRIDE)
        dz3t = torch.zeros(A, ldT, device='cuda')[:, :rows]
        vdz, vpart = f(rows), f(nbv, 8)
        if fused:
            K.epoch_losses(mode, mean, log_var, actions, behave, ref, adv, ctrl, gs, gk, part, True, True,
                           dz3, dlv, dq, st, dz3_t=dz3t, values=vals, returns=rets, v_dz3=vdz, v_partials=vpart)
        else:
            K.policy_loss(mode, mean, log_var, actions, behave, ref, adv, ctrl, gs, gk, part)
            K.policy_finalize(mode, part, nblk, gs, gk, log_var, rows, ctrl, True, True, dz3, dlv, dq, st, dz3_t=dz3t)
            K.value_loss(vals, rets, rows, vdz, vpart, ctrl, True)
        torch.cuda.synchronize()
        res.append([t.cpu().clone() for t in (gs, gk, part, dz3, dlv, dq, st, dz3t, vdz, vpart, ctrl.view(torch.int32))])
    for a, b in zip(*res):
        assert torch.equal(a, b)                 # same code, same reduction order: bit-identical
    assert int(res[1][-1][L.C_STEP_CRITIC]) == 1


def test_clip_adam_pair_equals_two_launches(K):
    g = torch.Generator().manual_seed(5)
    na, nc = 5000, 3333
    outs = []
    for pair in (False, True):
        ctrl = torch.zeros(L.CTRL_WORDS)
        ctrl[L.C_LR_ACTOR], ctrl[L.C_LR_CRITIC], ctrl[L.C_ACTOR_MAX_NORM], ctrl[L.C_CRITIC_MAX_NORM] = 1e-3, 2e-3, 5.0, 0.5
        ctrl = ctrl.cuda()
        ci = ctrl.view(torch.int32)
        ci[L.C_STEP_ACTOR], ci[L.C_STEP_CRITIC] = 3, 7
        gg = torch.Generator().manual_seed(9)
        ta, ga, tc, gc = (torch.randn(n, generator=gg).cuda() for n in (na, na, nc, nc))
        ma, va, mc, vc = (torch.zeros(n).cuda() for n in (na, na, nc, nc))
        pa, pc = torch.zeros(K.sumsq_blocks(na)).cuda(), torch.zeros(K.sumsq_blocks(nc)).cuda()
        K.sumsq_partials(ga, pa); K.sumsq_partials(gc, pc)
        gna, gnc = torch.zeros(1).cuda(), torch.zeros(1).cuda()
        if pair:
            K.clip_adam_pair((ta, ga, ma, va, pa, pa.numel(), True, gna), (tc, gc, mc, vc, pc, pc.numel(), False, gnc), ctrl)
        else:
            K.clip_adam(ta, ga, ma, va, pa, pa.numel(), ctrl, 0, True, gna)
            K.clip_adam(tc, gc, mc, vc, pc, pc.numel(), ctrl, 1, False, gnc)
        outs.append([t.cpu().clone() for t in (ta, ma, va, tc, mc, vc, gna, gnc)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('M,N,rows', [(400, 17, 7936), (400, 100, 31744), (16, 192, 102400), (33, 70, 5000), (8, 8, 900),
                                      (400, 100, 126976),     # the register-resident kernel (smx_wgrad.hip), one column group
                                      (400, 376, 40000),      # ... three column groups (128 + 128 + 120)
                                      (300, 288, 33000), (64, 48, 32768 + 7)])
def test_linear_wgrad_splitk_matches_fp64(K, M, N, rows):
    """weight gradients over many rows (LSTM: B*T, CNN: B*E*pixels): split-K partial tiles added in
    a fixed order; compared with an fp64 reference (both the split and the plain kernel are fp32
    sums, in different orders)"""
    g = torch.Generator().manual_seed(M + rows)
    dz, x = torch.randn(rows, M, generator=g), torch.randn(rows, N, generator=g)
    ref_w = (dz.double().t() @ x.double())
    ref_b = dz.double().sum(0)
    n_ws = K.linear_wgrad_ws_floats(M, N, rows)
    assert (n_ws > 0) == (rows >= 2048)
    ws = torch.empty(max(n_ws, 1), device='cuda')
    dW, db = torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')
    K.linear_wgrad(dz.cuda(), x.cuda(), dW, db, M, N, rows, ws=ws if n_ws else None)
    scale = float(ref_w.abs().max())
    close(dW.cpu().double() / scale, ref_w / scale, atol=2e-6, rtol=1e-5, msg='dW')
    close(db.cpu().double() / scale, ref_b / scale, atol=2e-6, rtol=1e-5, msg='db')
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    K.linear_wgrad(dz.cuda(), x.cuda(), dW2, db2, M, N, rows, ws=ws if n_ws else None)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)          # deterministic


@pytest.mark.parametrize('M,N1,N2,rows', [(400, 17, 100, 7936),       # the LSTM's dW_ih / dW_hh at 64 x 124 rows: ONE launch
                                          (400, 376, 100, 7936), (400, 17, 100, 126976),     # ... a kernel of its own: two
                                          (400, 17, 100, 900), (36, 5, 70, 4100)])
def test_linear_wgrad_pair_is_the_two_single_calls_bit_for_bit(K, M, N1, N2, rows):
    """smx_linear_wgrad_splitk_pair_f32: two weight gradients from the same dZ in one split-K launch where both run on the
    32 x 32 kernel -- every tile is the same workgroup program on the same chunks, so the results are those of two
    smx_linear_wgrad_splitk_f32 calls to the last bit (and it IS those two calls where a shape has another kernel)"""
    g = torch.Generator().manual_seed(M + rows + N1)
    dz = torch.randn(rows, M, generator=g).cuda()
    x1, x2 = torch.randn(rows, N1, generator=g).cuda(), torch.randn(rows, N2, generator=g).cuda()
    n1, n2 = K.linear_wgrad_ws_floats(M, N1, rows), K.linear_wgrad_ws_floats(M, N2, rows)
    ws = torch.empty(max(n1 + n2, 1), device='cuda')
    single = []
    for x, N, n in ((x1, N1, n1), (x2, N2, n2)):
        dW, db = torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')
        K.linear_wgrad(dz, x, dW, db, M, N, rows, ws=ws[:n] if n else None)
        single += [dW, db]
    for have_ws in (True, False):
        out = [torch.full((M, N1), 7.0, device='cuda'), torch.full((M,), 7.0, device='cuda'),
               torch.full((M, N2), 7.0, device='cuda'), torch.full((M,), 7.0, device='cuda')]
        K.linear_wgrad_pair(dz, x1, out[0], out[1], x2, out[2], out[3], rows, ws if have_ws and n1 + n2 else None)
        if have_ws:
            for a, b in zip(out, single):
                assert torch.equal(a, b)
        else:                       # no workspace: the plain kernels (other summation order)
            for a, b in zip(out, single):
                close(a.cpu().double(), b.cpu().double(), atol=1e-4 * rows ** 0.5, rtol=1e-4, msg='pair without a workspace')


@pytest.mark.parametrize('mode', [L.SMX_PPO_CLIP, L.SMX_PPO_ADAPT])
@pytest.mark.parametrize('rows,A,world,kl_target', [(1024, 17, 8, 1e9), (100, 6, 2, 1e9), (100, 6, 2, 1e-4),
                                                     (5, 2, 1, 1e9)])
def test_data_parallel_epoch_losses_and_combine(K, mode, rows, A, world, kl_target):
    """smx_ppo_epoch_losses_dp_f32 + smx_ppo_epoch_combine_f32 (one collective per epoch on several
    ranks) against the finalize path: with G_surr / G_kl the two right-hand sides pushed through a
    LINEAR map, combine(G_surr, G_kl) must equal that map applied to finalize's dz3."""
    from cpu_kernels import TorchCpuKernels
    C = TorchCpuKernels()
    g = torch.Generator().manual_seed(rows * 3 + A)
    n_total = rows * world
    log_var = torch.full((A,), -1.0) + 0.1 * torch.randn(A, generator=g)
    mean = torch.tanh(0.1 * torch.randn(rows, A, generator=g))
    std = torch.exp(log_var)
    actions = mean + std * torch.randn(rows, A, generator=g)
    behave = torch.cat([mean + 0.05 * torch.randn(rows, A, generator=g), (std * 1.1).expand(rows, A)], 1).contiguous()
    ref = torch.cat([mean + 0.2 * torch.randn(rows, A, generator=g), (std * 0.9).expand(rows, A)], 1).contiguous()
    adv = torch.randn(rows, generator=g)
    vals, rets = torch.randn(rows, generator=g) * 2, torch.randn(rows, generator=g) * 5
    n_mlp, n_c = 4100, 9000
    proj = torch.randn(rows * A, n_mlp, generator=g) / 30.0          # the "backward pass": linear in dz3
    grads_c0 = torch.randn(n_c, generator=g)
    nblk, nbv, stride = K.loss_blocks(rows), K.value_loss_blocks(rows), 8 + 2 * A
    ldT = rows + 16

    def ctrl_():
        c = torch.zeros(L.CTRL_WORDS)
        c[L.C_BETA], c[L.C_ETA], c[L.C_CLIP_EPS], c[L.C_KL_TARGET] = 1.0, 250.0, 0.2, kl_target
        return c
    outs = {}
    for name, KK, dev in (('hip', K, 'cuda'), ('cpu', C, 'cpu')):
        t = lambda x: x.to(dev)  # noqa: E731
        f = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
        ctrl = t(ctrl_())
        gs, gk, part = f(rows, A), f(rows, A), f(nblk + 2, stride)       # 2 rows no rank fills
        gst, gkt = f(A, ldT)[:, :rows], f(A, ldT)[:, :rows]
        vdz, vpart = f(rows), f(nbv, 8)
        KK.epoch_losses_dp(mode, t(mean), t(log_var), t(actions), t(behave), t(ref), t(adv), ctrl, gs, gk,
                           part, n_total, g_surr_t=gst, g_kl_t=gkt, values=t(vals), returns=t(rets),
                           v_dz3=vdz, v_partials=vpart)
        if name == 'hip':
            assert torch.equal(gst.cpu(), gs.cpu().t()) and torch.equal(gkt.cpu(), gk.cpu().t())
        # "all-reduce" over `world` identical ranks
        G_s, G_k = (gs.cpu().reshape(-1) @ proj) * world, (gk.cpu().reshape(-1) @ proj) * world
        part_ar = (part * world).contiguous()
        n_a = n_mlp + A + 3
        ga = t(torch.cat([G_s, torch.full((A,), 7.0), torch.zeros(3)]))
        gkl = t(G_k.clone()) if mode == L.SMX_PPO_ADAPT else None
        gc = t(grads_c0)
        sq_a, sq_c, st = f(KK.sumsq_blocks(n_a)), f(KK.sumsq_blocks(n_c)), f(L.PS_STRIDE)
        KK.epoch_combine(mode, part_ar, nblk + 2, n_total, t(log_var), ctrl, True, True, st, ga, gkl, n_mlp,
                         sq_a, gc, sq_c)
        if dev == 'cuda':
            torch.cuda.synchronize()
        outs[name] = dict(gs=gs.cpu(), gk=gk.cpu(), part=part.cpu(), vdz=vdz.cpu(), vpart=vpart.cpu(),
                          ga=ga.cpu(), st=st.cpu(), sq_a=float(sq_a.sum()), sq_c=float(sq_c.sum()),
                          ci=ctrl.view(torch.int32).cpu().clone())
    h, c = outs['hip'], outs['cpu']
    for k in ('gs', 'gk', 'part', 'vdz', 'vpart', 'ga', 'st'):
        # sums of terms that cancel: the absolute tolerance scales with the largest entry
        scale = max(1.0, float(c[k].abs().max()))
        np.testing.assert_allclose(h[k].numpy(), c[k].numpy(), rtol=2e-5, atol=2e-6 * scale, err_msg=k)
    assert torch.equal(h['ci'], c['ci']) and int(h['ci'][L.C_STEP_CRITIC]) == 1
    np.testing.assert_allclose(h['sq_c'], c['sq_c'], rtol=1e-5)
    stopped = int(h['ci'][L.C_STOP]) != 0
    assert stopped == (kl_target < 1.0)
    if stopped:          # the early exit: statistics are reported, no gradient is formed
        return
    np.testing.assert_allclose(h['sq_a'], c['sq_a'], rtol=1e-5)
    # and the finalize path pushed through the same linear map gives the same gradient
    f = lambda *s: torch.zeros(*s, device='cuda')  # noqa: E731
    ctrl = ctrl_().cuda()
    gs, gk, part, dz3, dlv, dq, st = f(rows, A), f(rows, A), f(nblk, stride), f(rows, A), f(A), f(1), f(L.PS_STRIDE)
    K.policy_loss(mode, mean.cuda(), log_var.cuda(), actions.cuda(), behave.cuda(), ref.cuda(), adv.cuda(), ctrl,
                  gs, gk, part)
    part_ar = (part * world).contiguous()
    K.policy_finalize(mode, part_ar, nblk, gs, gk, log_var.cuda(), n_total, ctrl, True, True, dz3, dlv, dq, st)
    torch.cuda.synchronize()
    want = (dz3.cpu().reshape(-1).double() @ proj.double()) * world
    np.testing.assert_allclose(h['ga'][:n_mlp].numpy(), want.numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(h['ga'][n_mlp:n_mlp + A].numpy(), dlv.cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert torch.equal(h['ga'][n_mlp + A:], torch.zeros(3))
    np.testing.assert_allclose(h['st'].numpy(), st.cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_acting_kernels(K):
    """z-filter straight from the running sums == stats + forward (bit-identical); the sampling head
    against the ATen expression of PPOAgent.act (ppo_agent.py:137-147), incl. row-strided outputs"""
    g = torch.Generator().manual_seed(3)
    n, D, A = 1000, 37, 5
    x = torch.randn(n, 3, D, generator=g).cuda()
    rs, rsq = (torch.randn(D, generator=g) * 50).cuda(), (torch.rand(D, generator=g) * 900 + 100).cuda()
    rsq[3] = 1e-9                                   # var < 0 -> NaN std -> NaN output, like torch
    rsq[4] = rs[4] * rs[4] / 40.0                   # var == 0 -> std clamps to eps
    cnt = torch.tensor([40.0]).cuda()
    mean, std, a, b = (torch.empty(D).cuda(), torch.empty(D).cuda(), torch.empty(n, D).cuda(),
                       torch.empty(n, D).cuda())
    K.zfilter_stats(rs, rsq, cnt, 1e-5, mean, std)
    K.zfilter_forward(x[:, 1], mean, std, a)
    K.zfilter_forward_sums(x[:, 1], rs, rsq, cnt, 1e-5, b)
    torch.cuda.synchronize()
    assert torch.equal(a.cpu().nan_to_num(nan=7.0), b.cpu().nan_to_num(nan=7.0))
    assert bool(torch.isnan(b[:, 3]).all()) and not bool(torch.isnan(b[:, :3]).any())

    mu = torch.tanh(torch.randn(n, 2 * A, generator=g))[:, :A].cuda()        # row stride 2A
    log_var = (torch.randn(A, generator=g) * 0.3 - 1).cuda()
    noise = torch.exp(torch.rand(n, generator=g) - 0.5).cuda()
    eps = torch.randn(n, A, generator=g).cuda() * 2
    roll = torch.zeros(n, 4, 2 * A).cuda()
    acts = torch.zeros(n, A).cuda()
    K.diaggauss_sample(mu, log_var, noise, eps, acts, roll[:, 2])
    pd = torch.cat([mu, torch.exp(log_var) * torch.ones_like(mu)], 1)
    pd[:, A:] *= noise.view(-1, 1)
    want = torch.clamp(eps * pd[:, A:] + pd[:, :A], -1, 1)
    torch.cuda.synchronize()
    np.testing.assert_allclose(roll[:, 2].cpu().numpy(), pd.cpu().numpy(), rtol=2e-6, atol=0)
    np.testing.assert_allclose(acts.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6)
    assert float(roll[:, [0, 1, 3]].abs().max()) == 0.0
    K.diaggauss_sample(mu, log_var, None, None, acts, None)                   # deterministic evaluation
    torch.cuda.synchronize()
    assert torch.equal(acts.cpu(), torch.clamp(mu, -1, 1).cpu())


def test_fused_rollout_equals_act_batch_loop_on_hip(K):
    """SyntheticVecEnv.rollout (three launches per step: two hidden layers + smx_synth_act_env_step_head_f32) against
    the per-step loop (act_batch: smx_zfilter_forward_sums_f32 + smx_mlp3_forward_f32 + smx_diaggauss_sample_f32, then
    smx_synth_env_step_f32) over a whole recorded rollout.  fp32 tolerance: the folded output layer sums its H2
    products in another order than the MFMA tiles; everything downstream of the mean is the same expression."""
    import test_hostpath as TH
    fused, loop = TH._rollout_pair()
    for k in fused:
        np.testing.assert_allclose(fused[k].numpy(), loop[k].numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    assert float(fused['pds'].abs().sum()) > 0


def test_act_env_step_head_equals_separate_launches(K):
    """smx_synth_act_env_step_head_f32 == output layer (smx_linear_f32 + tanh) + smx_synth_act_env_step_f32 on one
    step, incl. a partial last workgroup, deterministic mode and the episode-end reset"""
    g = torch.Generator().manual_seed(5)
    n, D, A, H2, T = 37, 11, 3, 24, 4
    for eps_on, t, ep in ((True, 0, 9), (False, 2, 3)):
        W3, b3 = torch.randn(A, H2, generator=g) / 5, torch.randn(A, generator=g) / 5
        h2 = torch.relu(torch.randn(n, H2, generator=g))
        state0, init = torch.randn(n, D, generator=g), torch.randn(n, D, generator=g)
        log_var = torch.randn(A, generator=g) * 0.3 - 1
        noise = torch.exp(torch.randn(n, generator=g) * 0.1)
        eps = torch.randn(n, A, generator=g) if eps_on else None

        class ZF:
            running_sum = dev(torch.randn(D, generator=g) * 3)
            running_sumsq = dev(torch.rand(D, generator=g) * 40 + 20)
            count = dev(torch.tensor([10.0]))
            eps = 1e-2
        outs = []
        for head in (True, False):
            state = dev(state0.clone())
            rolls = {k: torch.zeros(n, T, w).cuda() for k, w in (('obs', D), ('actions', A), ('pds', 2 * A))}
            rolls['rewards'], rolls['dones'] = torch.zeros(n, T).cuda(), torch.zeros(n, T).cuda()
            xn = torch.empty(n, D).cuda()
            if head:
                K.synth_act_env_step_head(dev(W3), dev(b3), dev(h2), L.SMX_ACT_TANH, state, dev(init), dev(log_var),
                                          dev(noise), dev(eps) if eps_on else None, t, ep, 1, rolls, ZF, xn)
            else:
                mean = torch.empty(n, A).cuda()
                K.linear(dev(h2), 1, dev(W3), 1, dev(b3), mean, n, A, H2, act=L.SMX_ACT_TANH)
                K.synth_act_env_step(state, dev(init), mean, dev(log_var), dev(noise), dev(eps) if eps_on else None,
                                     t, ep, 1, rolls, ZF, xn)
            outs.append(dict(rolls, state=state, xn=xn))
        for k in outs[0]:
            np.testing.assert_allclose(outs[0][k].cpu().numpy(), outs[1][k].cpu().numpy(), rtol=1e-5, atol=1e-6,
                                       err_msg=k)


def test_linear_tile_kernel_matches_fp64_and_the_rows_kernel_bit_for_bit(K, tmp_path):
    """the LDS-tiled throughput GEMM (gemm_tile_kernel: M >= 2048 rows, N >= 64, K >= 32; all four operand storage
    combinations, three tile shapes, ragged M / N / K, bias + ReLU / tanh / ReLU-mask epilogues) against a float64 GEMM,
    and against gemm_rows_kernel run in a subprocess (SMX_GEMM_ROWS_ONLY=1): same fragment mapping and K order, so the
    SAME BITS -- the host picks either by shape without moving a golden"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'diag'))
    import gemm_tile_cases as GC
    outs = GC.run_cases(K)
    for (M, N, Kd, akc, bkc, bias, act, mask), (A, B, b, mk, C) in zip(GC.CASES, outs):
        assert not bool(torch.isnan(C).any()), (M, N, Kd)
        rows = torch.cat([torch.arange(0, 160), torch.arange(M // 2, M // 2 + 160), torch.arange(M - 160, M)]).cuda()
        Ad = (A[rows] if akc else A[:, rows].t()).double()
        Bd = (B if bkc else B.t()).double()
        want = Ad @ Bd.t()
        if b is not None:
            want = want + b.double()
        want = torch.relu(want) if act == 1 else torch.tanh(want) if act == 2 else want
        if mk is not None:
            want = want * mk[rows].double()
        np.testing.assert_allclose(C[rows].cpu().numpy(), want.float().cpu().numpy(), rtol=2e-5, atol=2e-5,
                                   err_msg=str((M, N, Kd, akc, bkc)))
    ref = str(tmp_path / 'rows_kernel.npz')
    env = dict(os.environ, SMX_GEMM_ROWS_ONLY='1')
    subprocess.run([sys.executable, GC.__file__, ref], check=True, env=env, timeout=600)
    got = np.load(ref)
    for ci, o in enumerate(outs):
        assert np.array_equal(o[-1].cpu().numpy(), got['c%d' % ci]), 'case %d: tile and rows kernels differ' % ci


def test_linear_cuts_operands_past_2gib_into_row_blocks(K):
    """smx_linear_f32 addresses operands through 31-bit buffer descriptors; a row-major A past 2 GiB
    (a convolution's patch matrix over thousands of frames) becomes several problems of one launch.
    Rows around every cut and at both ends against torch's fp32 GEMM, incl. bias + ReLU + mask."""
    M, Kd, N = 3_000_000, 192, 16                       # A = 2.3 GB -> two row blocks
    g = torch.Generator(device='cuda').manual_seed(0)
    A = torch.randn(M, Kd, device='cuda', generator=g)
    W = torch.randn(N, Kd, device='cuda', generator=g) / 8
    b = torch.randn(N, device='cuda', generator=g)
    C = torch.full((M, N), float('nan'), device='cuda')
    K.linear(A, 1, W, 1, b, C, M, N, Kd, act=L.SMX_ACT_RELU)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(C).any())
    idx = torch.cat([torch.arange(0, 300), torch.arange(M // 2 - 4000, M // 2 + 4000), torch.arange(M - 300, M)]).cuda()
    want = torch.relu(A[idx].double() @ W.double().t() + b.double())
    np.testing.assert_allclose(C[idx].cpu().numpy(), want.float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    # the data-gradient form: no bias, ReLU mask
    mask = (torch.rand(M, N, device='cuda', generator=g) > 0.5).float()
    K.linear(A, 1, W, 1, None, C, M, N, Kd, relu_mask=mask)
    torch.cuda.synchronize()
    want = (A[idx].double() @ W.double().t()) * mask[idx].double()
    np.testing.assert_allclose(C[idx].cpu().numpy(), want.float().cpu().numpy(), rtol=2e-5, atol=2e-5)


def test_linear_wgrad_splitk_over_an_operand_past_2gib(K):
    """conv1's weight gradient at cfg 4 scale: dW = dZ^T . X over 3e6 patch rows, X = 2.3 GB (each
    split-K chunk addresses its own slice); fp64 reference on the device"""
    M, N, rows = 16, 192, 3_000_000
    g = torch.Generator(device='cuda').manual_seed(1)
    dz = torch.randn(rows, M, device='cuda', generator=g)
    x = torch.randn(rows, N, device='cuda', generator=g)
    ref_w, ref_b = torch.zeros(M, N, dtype=torch.float64, device='cuda'), dz.double().sum(0)
    for r0 in range(0, rows, 500_000):                     # bounded fp64 temporaries
        ref_w += dz[r0:r0 + 500_000].double().t() @ x[r0:r0 + 500_000].double()
    ws = torch.empty(K.linear_wgrad_ws_floats(M, N, rows), device='cuda')
    dW, db = torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')
    K.linear_wgrad(dz, x, dW, db, M, N, rows, ws=ws)
    torch.cuda.synchronize()
    scale = float(ref_w.abs().max())
    close(dW.cpu().double() / scale, ref_w.cpu() / scale, atol=3e-6, rtol=1e-5, msg='dW')
    close(db.cpu().double() / scale, ref_b.cpu() / scale, atol=3e-6, rtol=1e-5, msg='db')


def test_window_cut_fifo_round_trip_at_baseline_size(K):
    """BASELINE cfg 5 (1024 actors x 128 steps x 376): the moving-window cut of a whole rollout, the
    FIFO insert (through the table and zero-copy) and the pop are byte copies -- the popped batch must
    be the rollout's slices bit for bit; cfg 3: a uniform sample of 512 out of 1e6 rows returns
    exactly the rows its indices name"""
    from surreal_amd.env import SyntheticVecEnv
    from surreal_amd.replay import FIFOReplay, UniformReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    n, T, D, A = 1024, 128, 376, 17
    venv = SyntheticVecEnv(n, D, A, episode_len=T)
    venv.start_rollout(T, info_width=2 * A)
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, r in venv.rolls.items():
        r.copy_(torch.randn(r.shape, device='cuda', generator=g))
    venv.slot = T
    lc = ppo_learner_config()
    lc.algo.n_step, lc.replay.batch_size, lc.replay.memory_size = T, n, 2 * n
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/x')
    for zero_copy in (False, True):
        f = FIFOReplay(lc, ec, sc)
        slots = f.reserve_batch(n, venv.window_shapes(T)) if zero_copy else None
        if zero_copy:
            assert slots is not None
            venv.emit_windows(T, T, out=slots)
            f.commit_batch(n)
        else:
            f.insert_batch(venv.emit_windows(T, T))
        b = f.sample_batch(n, copy=not zero_copy)
        torch.cuda.synchronize()
        assert torch.equal(b['obs'], venv.rolls['obs'][:, :T]) and torch.equal(b['obs_next'][:, 0], venv.rolls['obs'][:, T])
        assert torch.equal(b['actions'], venv.rolls['actions'][:, :T]) and torch.equal(b['pds'], venv.rolls['pds'][:, :T])
        assert torch.equal(b['rewards'], venv.rolls['rewards'][:, :T]) and torch.equal(b['dones'], venv.rolls['dones'][:, :T])
        assert len(f) == 0
    lc.replay.memory_size, lc.replay.batch_size = 1000000, 512
    u = UniformReplay(lc, ec, sc)
    rows = torch.randn(1000000, 17, device='cuda', generator=g)
    u.insert_batch({'obs': rows, 'rewards': rows[:, 0].contiguous()})
    idx = u.sample_indices(512)
    assert int(idx.min()) >= 0 and int(idx.max()) < 1000000 and len(torch.unique(idx)) > 500
    got = u.sample_batch(512, indices=idx)
    assert torch.equal(got['obs'], rows[idx]) and torch.equal(got['rewards'], rows[idx, 0])


@pytest.mark.parametrize('n', [7, 1024 * 128, 70001])
def test_reward_filter_scale_forward_update(K, n):
    """smx_reward_filter_f32 against RewardFilter.forward / .update written with the reference's own torch ops
    (surreal/model/reward_filter.py:33-57): three consecutive batches (the second call filters with the state
    the first left behind, incl. the running_sumsq ASSIGNMENT), scale-only mode, the sums-only mode several
    ranks use, and in-place output"""
    g = torch.Generator().manual_seed(n)
    state = torch.tensor([1e-5, 0.0, 0.0])
    dstate = state.clone().cuda()
    part = torch.zeros(K.reward_filter_partials(), dtype=torch.float64, device='cuda')
    ticket = torch.zeros(1, dtype=torch.int32, device='cuda')
    for it in range(3):
        r = torch.randn(n, generator=g) * (1.0 + it) + 0.3 * it
        x = r * 0.25
        mean = state[1] / state[0]
        std = torch.clamp((state[2] / state[0] - mean.pow(2)).pow(0.5), min=1e-5)
        want = torch.clamp((x - mean) / std, -5.0, 5.0)
        state = torch.stack([state[0] + float(n), state[1] + x.sum(), (x * x).sum()])
        out = torch.empty(n, device='cuda')
        K.reward_filter(r.cuda(), 0.25, dstate, 1e-5, out, part, ticket)
        close(out, want, msg='filtered rewards, batch %d' % it)
        close(dstate, state, rtol=2e-6, atol=1e-3, msg='filter state after batch %d' % it)
        assert int(ticket.item()) == 0
    # scale only: no filter, state untouched; in place
    r = torch.randn(n, generator=g)
    buf = r.clone().cuda()
    before = dstate.clone()
    K.reward_filter(buf, 3.0, dstate, 1e-5, buf, part, ticket, use_filter=False, update=False)
    close(buf, r * 3.0, atol=0, rtol=0)
    assert torch.equal(before, dstate)
    # what several ranks run: filter with the current state, report the batch sums, leave the state alone
    sums = torch.zeros(3, device='cuda')
    out = torch.empty(n, device='cuda')
    K.reward_filter(r.cuda(), 0.5, dstate, 1e-5, out, part, ticket, use_filter=True, update=False, sums=sums)
    x = r * 0.5
    close(sums, torch.stack([torch.tensor(float(n)), x.sum(), (x * x).sum()]), rtol=2e-6, atol=1e-3)
    assert torch.equal(before, dstate)


def _rollout_setup(n, D, A, hidden, T, episode_len, use_z, deterministic, seed):
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticVecEnv
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    from surreal_amd import synthetic
    lc = ppo_learner_config()
    lc.algo.rnn.if_rnn_policy = False
    lc.algo.use_z_filter = use_z
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = list(hidden)
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_rollout')
    agent = PPOAgent(lc, ec, sc, agent_id=1, agent_mode='eval_deterministic_local' if deterministic else 'training')
    agent.model.load_params(synthetic.make_ppo_params(D, A, hidden=tuple(hidden), seed=seed, final_scale=2.0,
                                                      log_sig_spread=0.4))
    if use_z:
        agent.model.z_filter.load_state_dict(synthetic.make_zfilter_state(D, seed=seed + 1))
    eps = None if deterministic else torch.randn(T, n, A, generator=torch.Generator().manual_seed(seed)).cuda()

    def run(how):
        venv = SyntheticVecEnv(n, D, A, episode_len=episode_len, seeds=list(range(n)))
        # (an episode may end inside the recorded span here: the kernels' reset path is exercised on purpose)
        venv.T, venv.slot = T, 0
        f = lambda *s: torch.zeros(*s, device='cuda')  # noqa: E731
        venv.rolls = {'obs': f(n, T + 1, D), 'actions': f(n, T + 1, A), 'rewards': f(n, T + 1), 'dones': f(n, T + 1),
                      'pds': f(n, T + 1, 2 * A)}
        if how == 'persistent':
            venv.persistent = True
            venv.rollout(agent, eps=eps)
        elif how == 'reference':
            venv.rollout_reference(agent, eps)
        else:
            venv.persistent = False
            venv.rollout(agent, eps=eps)
        torch.cuda.synchronize()
        assert venv.slot == T
        out = {k: v.cpu() for k, v in venv.rolls.items()}
        out['state'], out['t'] = venv.state.cpu(), venv.t
        return out
    return agent, run


@pytest.mark.parametrize('n,D,A,hidden,T,ep,use_z,det', [
    (37, 11, 3, (24, 16), 9, 9, True, False),          # a partial last workgroup (37 = 2 x 16 + 5)
    (16, 376, 17, (300, 200), 6, 50, True, False),     # the benchmark's policy shape
    (48, 17, 6, (300, 200), 7, 4, True, False),        # the episode ends (and resets) inside the rollout
    (20, 29, 5, (40, 24), 5, 5, False, False),         # no z-filter
    (33, 12, 2, (16, 12), 4, 9, True, True),           # deterministic mode: no draws
])
def test_persistent_rollout_kernel(K, n, D, A, hidden, T, ep, use_z, det):
    """smx_synth_rollout_f32 (one launch, 4 / 8 / 16 actors per workgroup through all T steps on the 4-row MFMA loop,
    smx_rows4_mma.inc.h) records what the two-launches-per-step loop (smx_epoch_forward_f32 -- the 16-row loop -- for
    the means, then smx_synth_act_env_step_f32) and the layered per-step path (GEMM launches per layer) record, to fp32
    rounding of the layer sums: the three sum a layer's products in three different orders.  Everything that is not a
    function of the means is exact: dones, the step counter, the zero pattern.  Every row-group count is run (the host
    picks 1 for <= 1024 actors; SMX_ROLLOUT_RG forces it)."""
    agent, run = _rollout_setup(n, D, A, hidden, T, ep, use_z, det, seed=11)
    assert K.synth_rollout_supported(agent.model.actor)
    ref, layered = run('reference'), run('layered')
    one = run('persistent')
    for k in ref:
        if k == 't':
            assert one[k] == ref[k] == layered[k]
            continue
        if k == 'dones':
            assert torch.equal(one[k], ref[k])
        np.testing.assert_allclose(one[k].numpy(), ref[k].numpy(), rtol=2e-6, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(one[k].numpy(), layered[k].numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    assert float(one['pds'].abs().sum()) > 0 and float(one['obs'][:, T].abs().sum()) > 0
    if ep < T:
        assert float(one['dones'][:, :T].sum()) == n * (T // ep)


@pytest.mark.parametrize('rg', [1, 2, 4])
def test_persistent_rollout_kernel_row_group_counts_agree(K, rg):
    """4 and 8 actors per workgroup are the same arithmetic per actor (a row group is independent of its neighbours in
    the workgroup): bit-identical recordings, incl. a partial last workgroup.  16 actors per workgroup (what more than
    2048 actors run on) is the 16x16x4 loop: equal to fp32 rounding of the layer sums."""
    import subprocess
    import sys
    # the row-group override is read once per process: one child per forced value, compared through a file
    code = r'''
import sys, torch
sys.path.insert(0, %r)
import test_gpu_kernels as TK
agent, run = TK._rollout_setup(37, 29, 5, (40, 24), 6, 4, True, False, seed=13)
out = run('persistent')
torch.save({k: v for k, v in out.items()}, sys.argv[1])
''' % os.path.dirname(os.path.abspath(__file__))
    import tempfile
    outs = []
    for force in ('0', str(rg)):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            env = dict(os.environ, SMX_ROLLOUT_RG=force)
            r = subprocess.run([sys.executable, '-c', code, f.name], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-3000:]
            outs.append(torch.load(f.name))
    for k in outs[0]:
        if k == 't':
            assert outs[0][k] == outs[1][k]
        elif rg == 4:
            np.testing.assert_allclose(outs[0][k].numpy(), outs[1][k].numpy(), rtol=2e-6, atol=2e-6, err_msg=k)
        else:
            assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize('n,D,A,hidden,T', [(37, 11, 3, (24, 16), 9), (32, 376, 17, (300, 200), 6)])
def test_rollout_recorded_straight_into_the_replay_slots(K, n, D, A, hidden, T):
    """stride == n_step: a window IS the rollout (exp_sender_wrapper.py:209-228), so the one-launch rollout kernel
    records straight into the FIFO's reserved slots (SyntheticVecEnv.rollout_into: tables without the extra row, the
    observation after the last step into obs_next) -- bit for bit what rollout -> window cut -> insert leaves in the
    replay, and what the learner pops as views"""
    from surreal_amd.env import SyntheticVecEnv
    from surreal_amd.replay import FIFOReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    agent, _ = _rollout_setup(n, D, A, hidden, T, T, True, False, seed=21)
    eps = torch.randn(T, n, A, generator=torch.Generator().manual_seed(5)).cuda()
    lc = ppo_learner_config()
    lc.algo.n_step = lc.algo.stride = T
    lc.replay.batch_size, lc.replay.memory_size = n, 2 * n
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_rollout_into')

    def three_launch_path():
        venv = SyntheticVecEnv(n, D, A, episode_len=T, seeds=list(range(n)))
        venv.start_rollout(T, info_width=2 * A)
        venv.rollout(agent, eps=eps)
        replay = FIFOReplay(lc, ec, sc)
        replay.insert_batch(venv.emit_windows(T, T))
        return replay.sample_batch(n), venv.state.clone(), venv.t

    def zero_copy_path():
        venv = SyntheticVecEnv(n, D, A, episode_len=T, seeds=list(range(n)))
        venv.start_rollout(T, info_width=2 * A)            # (for window_shapes; its tables stay untouched)
        assert venv.can_rollout_into(agent)
        replay = FIFOReplay(lc, ec, sc)
        slots = replay.reserve_batch(n, venv.window_shapes(T))
        venv.rollout_into(agent, slots, eps=eps)
        replay.commit_batch(n)
        assert float(venv.rolls['obs'].abs().sum()) == 0.0
        return replay.sample_batch(n, copy=False), venv.state.clone(), venv.t
    (a, sa, ta), (b, sb, tb) = three_launch_path(), zero_copy_path()
    torch.cuda.synchronize()
    assert set(a) == set(b) and ta == tb and torch.equal(sa, sb)
    for k in a:
        assert torch.equal(a[k].reshape(b[k].shape), b[k]), k
    assert float(b['obs_next'].abs().sum()) > 0 and float(b['pds'].abs().sum()) > 0


def test_partials_fold_and_many_row_zupdate(K):
    """the two helpers of the many-row stem path: smx_ppo_partials_fold_f32 (thousands of loss partial rows -> 64) and
    smx_zfilter_update_ws_f32 (column sums over 126 976 rows by many workgroups) against float64 sums / the one-launch form"""
    g = torch.Generator(device='cuda').manual_seed(5)
    for nblk, stride, nout in ((7936, 20, 64), (7936, 42, 64), (300, 20, 64), (65, 9, 64)):
        part = torch.randn(nblk, stride, device='cuda', generator=g)
        out = torch.full((nout, stride), float('nan'), device='cuda')
        K.partials_fold(part, nblk, out)
        R = (nblk + nout - 1) // nout
        for j in (0, 1, nout // 2, nout - 1):
            want = part[j * R:min(nblk, (j + 1) * R)].double().sum(0)
            close(out[j], want.float(), atol=2e-5, rtol=1e-5, msg='fold row %d of %d x %d' % (j, nblk, stride))
        close(out.double().sum(0).float(), part.double().sum(0).float(), atol=1e-4, rtol=1e-5)
    ctrl = torch.zeros(64, dtype=torch.int32, device='cuda')
    ctrl[L.C_STOP] = 1
    out.fill_(3.0)
    K.partials_fold(part, nblk, out, ctrl.view(torch.float32))
    assert float(out.min()) == 3.0
    for rows, D in ((126976, 17), (40000, 376), (16384, 5), (9000, 17)):
        x = torch.randn(rows, D, device='cuda', generator=g) * 2 + 0.5
        nws = K.zfilter_update_ws_floats(rows, D)
        assert (nws > 0) == (rows >= 16384)
        ws = torch.empty(max(nws, 1), device='cuda')
        rs0, rq0, c0 = torch.randn(D, device='cuda', generator=g), torch.rand(D, device='cuda', generator=g) * 50, \
            torch.tensor([1000.0], device='cuda')
        rs1, rq1, c1 = rs0.clone(), rq0.clone(), c0.clone()
        rs2, rq2, c2 = rs0.clone(), rq0.clone(), c0.clone()
        K.zfilter_update(x, rs1, rq1, c1, rows, ws=ws)
        K.zfilter_update(x, rs2, rq2, c2, rows)
        want_s, want_q = rs0.double() + x.double().sum(0), rq0.double() + (x.double() ** 2).sum(0)
        close(rs1, want_s.float(), atol=1e-2, rtol=2e-6, msg='sum %d x %d' % (rows, D))
        close(rq1, want_q.float(), atol=1e-1, rtol=2e-6, msg='sumsq')
        close(rs1, rs2, atol=1e-2, rtol=2e-6), close(rq1, rq2, atol=1e-1, rtol=2e-6)
        assert float(c1) == float(c2) == 1000.0 + rows
        if nws == 0:
            assert torch.equal(rs1, rs2) and torch.equal(rq1, rq2)


@pytest.mark.parametrize('rows,F', [(512, 400), (37, 24), (16, 300), (5, 1000), (130, 72)])
def test_layernorm_forward_backward_match_aten(K, rows, F):
    """smx_layernorm_forward_f32 / _backward_f32 (DDPG use_layernorm) against torch.nn.functional.layer_norm + autograd on
    the CPU, rows of x / y / dy strided (views into wider buffers), with and without the ReLU mask of the layer in front"""
    g = torch.Generator().manual_seed(rows + F)
    wide = torch.randn(rows, F + 3, generator=g)
    x = torch.relu(wide[:, :F]).clone().requires_grad_(True)                  # the LayerNorm sits behind a ReLU
    gamma = (1 + 0.2 * torch.randn(F, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(F, generator=g)).requires_grad_(True)
    y = torch.nn.functional.layer_norm(x, (F,), gamma, beta, 1e-5)
    dy = torch.randn(rows, F, generator=g)
    (y * dy).sum().backward()
    xd = torch.zeros(rows, F + 3, device='cuda')
    xd[:, :F] = x.detach().cuda()
    yd = torch.full((rows, F + 5), 7.0, device='cuda')
    mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    K.layernorm_forward(xd[:, :F], gamma.detach().cuda(), beta.detach().cuda(), 1e-5, yd[:, :F], mean, rstd)
    close(yd[:, :F], y.detach(), msg='y')
    assert float(yd[:, F:].min()) == 7.0
    close(mean, x.detach().mean(1), msg='mean')
    dyd = torch.zeros(rows, F + 2, device='cuda')
    dyd[:, :F] = dy.cuda()
    ws = torch.empty(K.layernorm_backward_ws_floats(rows, F), device='cuda')
    for mask in (False, True):
        dx, dg, db = torch.full((rows, F + 1), 3.0, device='cuda'), torch.empty(F, device='cuda'), torch.empty(F, device='cuda')
        K.layernorm_backward(dyd[:, :F], xd[:, :F], mean, rstd, gamma.detach().cuda(), dx[:, :F], dg, db, ws, relu_mask=mask)
        want = x.grad * (x.detach() > 0) if mask else x.grad
        sc = float(want.abs().max())
        close(dx[:, :F] / sc, want / sc, atol=2e-6, rtol=2e-5, msg='dx mask=%s' % mask)
        assert float(dx[:, F:].min()) == 3.0
        sg = float(gamma.grad.abs().max())
        close(dg / sg, gamma.grad / sg, atol=2e-6, rtol=2e-5, msg='dgamma')
        close(db / sg, beta.grad / sg, atol=2e-6, rtol=2e-5, msg='dbeta')
