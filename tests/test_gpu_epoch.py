"""GPU tier (-m gpu): the fused row-block epoch kernels (csrc/smx_epoch.hip: forward + loss,
finalize + data gradients; smx_mlp3_wgrad_multi_f32) through the C ABI against the torch-CPU statement of
the same contract (tests/cpu_kernels.py) on identical seeded inputs -- shapes of the benchmark
(cfg 5), cfg 2, ragged row counts (partial last row block), widths that are not tile multiples, both
PPO modes, the KL early exit and the forward-only final pass.  1e-5 abs + rel (fp32)."""
import collections

import numpy as np
import pytest
import torch

from surreal_amd import _lib as L
from cpu_kernels import TorchCpuKernels
from test_gpu_kernels import make_net, close, dev

pytestmark = pytest.mark.gpu
C = TorchCpuKernels()


@pytest.fixture(scope='module')
def K():
    from surreal_amd.kernels import HipKernels
    return HipKernels()


def ctrl_block(beta=1.0, eta=250.0, eps=0.2, kl_target=0.015):
    c = torch.zeros(L.CTRL_WORDS)
    c[L.C_LR_ACTOR], c[L.C_LR_CRITIC] = 1e-4, 1e-4
    c[L.C_BETA], c[L.C_ETA], c[L.C_CLIP_EPS], c[L.C_KL_TARGET] = beta, eta, eps, kl_target
    c[L.C_ACTOR_MAX_NORM], c[L.C_CRITIC_MAX_NORM] = 5.0, 5.0
    return c


def build(rows, D, H1, H2, A, seed, mode, device):
    """one side (cpu or cuda) of the problem: identical values, device-local tensors"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    act_c, act_d = make_net(D, H1, H2, A, seed + 1, 'cuda')
    cri_c, cri_d = make_net(D, H1, H2, 1, seed + 2, 'cuda')
    x = r(rows, D)
    log_var = -1.0 + 0.2 * r(A)
    mean_b = 0.3 * torch.tanh(r(rows, A))
    sd = torch.exp(log_var).expand(rows, A)
    actions = torch.clamp(mean_b + sd * r(rows, A), -1, 1)
    behave = torch.cat([mean_b, sd * (1 + 0.05 * torch.rand(rows, A, generator=g))], 1)
    ref = torch.cat([0.3 * torch.tanh(r(rows, A)), sd.clone()], 1)
    adv = r(rows)
    returns = 2.0 * r(rows)
    ldT = rows + 16
    T = {}

    def side(dv, act, cri, Kx):
        f = lambda *s: torch.zeros(*s, device=dv)  # noqa: E731
        ft = lambda n: torch.zeros(n, ldT, device=dv)[:, :rows]  # noqa: E731
        nblk = (rows + 15) // 16
        t = dict(x=x.to(dv), log_var=log_var.to(dv), actions=actions.to(dv), behave=behave.to(dv), ref=ref.to(dv),
                 adv=adv.to(dv), returns=returns.to(dv), ctrl=ctrl_block().to(dv),
                 mean=f(rows, A), vpred=f(rows), g_surr=f(rows, A), g_kl=f(rows, A), partials=f(nblk, 8 + 2 * A),
                 v_dz3=f(rows), v_partials=f(nblk, 8), dlogvar=f(A), dlq=f(1), stats=f(L.PS_STRIDE),
                 xT=ft(D), h1aT=ft(H1), h2aT=ft(H2), h1cT=ft(H1), h2cT=ft(H2), dz3aT=ft(A), dz2aT=ft(H2),
                 dz1aT=ft(H1), dz2cT=ft(H2), dz1cT=ft(H1),
                 grads_a=f(act.numel), grads_c=f(cri.numel), act=act, cri=cri,
                 pk_a=f(max(Kx.epoch_packed_numel(act), act.numel)), pk_c=f(max(Kx.epoch_packed_numel(cri), cri.numel)))
        t['xT'].copy_(t['x'].t())
        return t
    from surreal_amd.kernels import HipKernels
    T['c'] = side('cpu', act_c, cri_c, C)
    T['d'] = side('cuda', act_d, cri_d, HipKernels())
    return T


def run(Kx, t, mode, check_stop=True, will_update=True, stop=None, phase='all'):
    rows = t['x'].shape[0]
    loss = dict(mode=mode, rows=rows, log_var=t['log_var'], actions=t['actions'], behave=t['behave'], ref=t['ref'],
                adv=t['adv'], g_surr=t['g_surr'], g_kl=t['g_kl'], partials=t['partials'], check_stop=check_stop,
                will_update=will_update, dlogvar=t['dlogvar'], dlogvar_sumsq=t['dlq'], stats=t['stats'],
                returns=t['returns'], v_dz3=t['v_dz3'], v_partials=t['v_partials'], v_will_update=True)
    if phase in ('all', 'fwd'):
        Kx.epoch_pack([(t['act'], t['pk_a']), (t['cri'], t['pk_c'])])
    aj = dict(net=t['act'], packed=t['pk_a'], x=t['x'], h1T=t['h1aT'], h2T=t['h2aT'], out=t['mean'], act=L.SMX_ACT_TANH, loss='policy',
              stop=stop, dz3T=t['dz3aT'], dz2T=t['dz2aT'], dz1T=t['dz1aT'], xT=t['xT'], grads=t['grads_a'])
    cj = dict(net=t['cri'], packed=t['pk_c'], x=t['x'], h1T=t['h1cT'], h2T=t['h2cT'], out=t['vpred'].view(-1, 1), act=L.SMX_ACT_NONE,
              loss='value', dz3=t['v_dz3'], dz3T=t['v_dz3'], dz2T=t['dz2cT'], dz1T=t['dz1cT'], xT=t['xT'],
              grads=t['grads_c'])
    if phase in ('all', 'fwd'):
        Kx.epoch_forward([aj, cj], loss, t['ctrl'], rows)
    if phase == 'fwd':
        return
    if will_update:
        Kx.epoch_backward([aj, cj], loss, t['ctrl'], rows)
        Kx.mlp3_wgrad_multi([aj, cj])
    else:
        Kx.epoch_backward([aj], loss, t['ctrl'], rows)


SHAPES = [(1024, 376, 300, 200, 17), (64, 17, 300, 200, 6), (8, 11, 24, 16, 3), (37, 29, 40, 24, 5), (37, 28, 40, 24, 5), (16, 12, 24, 16, 3),
          (100, 64, 332, 212, 32), (5, 8, 16, 12, 1)]


@pytest.mark.parametrize('mode', [L.SMX_PPO_ADAPT, L.SMX_PPO_CLIP])
@pytest.mark.parametrize('rows,D,H1,H2,A', SHAPES)
def test_epoch_forward_backward_wgrad(K, rows, D, H1, H2, A, mode):
    T = build(rows, D, H1, H2, A, seed=rows + D, mode=mode, device='cuda')
    run(C, T['c'], mode, phase='fwd')
    run(K, T['d'], mode, phase='fwd')
    torch.cuda.synchronize()
    c, d = T['c'], T['d']
    for k in ('mean', 'vpred', 'h1aT', 'h2aT', 'h1cT', 'h2cT'):
        close(d[k], c[k], msg=k)
    # the loss tiles and block sums (sums of 16 rows: tolerance scales with the magnitudes summed)
    for k in ('g_surr', 'g_kl', 'v_dz3'):
        close(d[k], c[k], atol=1e-5, rtol=2e-5, msg=k)
    close(d['partials'], c['partials'], atol=2e-4, rtol=2e-5, msg='policy partial rows')
    close(d['v_partials'], c['v_partials'], atol=2e-4, rtol=2e-5, msg='value partial rows')
    # the backward is compared like with like: a hidden unit whose pre-activation sits within
    # rounding of zero may be masked on one side and not on the other, so the CPU statement
    # continues from the device's own forward results
    for k in ('h1aT', 'h2aT', 'h1cT', 'h2cT', 'g_surr', 'g_kl', 'partials', 'v_dz3', 'v_partials'):
        c[k].copy_(d[k].cpu())
    run(C, T['c'], mode, phase='bwd')
    run(K, T['d'], mode, phase='bwd')
    torch.cuda.synchronize()
    close(d['stats'], c['stats'], atol=1e-5, rtol=2e-5, msg='stats')
    close(d['dlogvar'], c['dlogvar'], atol=1e-6, rtol=2e-5, msg='dlogvar')
    close(d['dlq'], c['dlq'], atol=1e-8, rtol=1e-4, msg='dlogvar sumsq')
    assert torch.equal(d['ctrl'].cpu().view(torch.int32)[L.C_STEP_ACTOR:L.C_EPOCHS_DONE + 1],
                       c['ctrl'].view(torch.int32)[L.C_STEP_ACTOR:L.C_EPOCHS_DONE + 1])
    for k in ('dz3aT', 'dz2aT', 'dz1aT', 'dz2cT', 'dz1cT'):
        close(d[k], c[k], atol=1e-7, rtol=2e-5, msg=k)
    for k in ('grads_a', 'grads_c'):
        close(d[k], c[k], atol=1e-6, rtol=5e-5, msg=k)


def _fb_pair(K, rows, D, H1, H2, A, mode, seed, kl_target=1e9, launches=1):
    """the same problem twice on the device: forward + backward as two launches / as smx_epoch_fwdbwd_f32"""
    out = []
    for fb in (False, True):
        t = build(rows, D, H1, H2, A, seed=seed, mode=mode, device='cuda')['d']
        if kl_target is not None:
            t['ctrl'][L.C_KL_TARGET] = kl_target
        t['sync'] = torch.zeros(4, dtype=torch.int32, device='cuda')
        t['slots'] = torch.zeros(4, 2 * ((rows + 15) // 16), dtype=torch.int32, device='cuda')
        loss = dict(mode=mode, rows=rows, log_var=t['log_var'], actions=t['actions'], behave=t['behave'], ref=t['ref'],
                    adv=t['adv'], g_surr=t['g_surr'], g_kl=t['g_kl'], partials=t['partials'], check_stop=True,
                    will_update=True, dlogvar=t['dlogvar'], dlogvar_sumsq=t['dlq'], stats=t['stats'],
                    returns=t['returns'], v_dz3=t['v_dz3'], v_partials=t['v_partials'], v_will_update=True)
        K.epoch_pack([(t['act'], t['pk_a']), (t['cri'], t['pk_c'])])
        aj = dict(net=t['act'], packed=t['pk_a'], x=t['x'], h1T=t['h1aT'], h2T=t['h2aT'], act=L.SMX_ACT_TANH, loss='policy',
                  dz3T=t['dz3aT'], dz2T=t['dz2aT'], dz1T=t['dz1aT'], xT=t['xT'], grads=t['grads_a'])
        cj = dict(net=t['cri'], packed=t['pk_c'], x=t['x'], h1T=t['h1cT'], h2T=t['h2cT'], act=L.SMX_ACT_NONE,
                  loss='value', dz3=t['v_dz3'], dz3T=t['v_dz3'], dz2T=t['dz2cT'], dz1T=t['dz1cT'], xT=t['xT'],
                  grads=t['grads_c'])
        for k in range(launches):
            if fb:
                K.epoch_fwdbwd([aj, cj], loss, t['ctrl'], rows, t['sync'][k:k + 1], t['slots'][k])
            else:
                K.epoch_forward([aj, cj], loss, t['ctrl'], rows)
                K.epoch_backward([aj, cj], loss, t['ctrl'], rows)
        K.mlp3_wgrad_multi([aj, cj])
        torch.cuda.synchronize()
        out.append(t)
    return out


# (4500 rows: 282 + 282 workgroups, more than the device has CUs -- in adapt mode the launch cannot keep every actor
# workgroup resident and runs as its two launches; in clip mode nothing waits and the one launch takes any size)
@pytest.mark.parametrize('mode', [L.SMX_PPO_ADAPT, L.SMX_PPO_CLIP])
@pytest.mark.parametrize('rows,D,H1,H2,A', SHAPES + [(4500, 20, 40, 24, 5)])
def test_epoch_fwdbwd_equals_forward_then_backward(K, rows, D, H1, H2, A, mode):
    """smx_epoch_fwdbwd_f32 (one launch, batch means through the in-launch counter) against the two launches it
    replaces, both on the device: the forward results and the loss sums bit for bit, the data gradients up to the
    rounding of where c_kl is applied (one layer later), the weight gradients that follow within 2e-5."""
    if not K.epoch_fwdbwd_supported(*[make_net(D, H1, H2, o, 1, 'cuda')[1] for o in (A, 1)]):
        pytest.skip('shape outside smx_epoch_fwdbwd_supported')
    # (no early exit here: in clip mode the fused launch forms the data gradients before workgroup 0 knows about it --
    # the optimiser launch honours the flag -- where the separate backward launch returns early)
    two, one = _fb_pair(K, rows, D, H1, H2, A, mode, seed=rows + D)
    for k in ('h1aT', 'h2aT', 'h1cT', 'h2cT', 'partials', 'v_dz3', 'v_partials'):    # (g_surr / g_kl stay in LDS)
        assert torch.equal(one[k], two[k]), k
    many = 2 * ((rows + 15) // 16) > torch.cuda.get_device_properties(0).multi_processor_count
    close(one['stats'], two['stats'], atol=1e-6, rtol=2e-6, msg='stats')
    close(one['dlogvar'], two['dlogvar'], atol=1e-7, rtol=2e-6, msg='dlogvar')
    close(one['dlq'], two['dlq'], atol=1e-9, rtol=1e-5, msg='dlogvar sumsq')
    assert torch.equal(one['ctrl'].view(torch.int32)[L.C_STEP_ACTOR:], two['ctrl'].view(torch.int32)[L.C_STEP_ACTOR:])
    assert int(one['sync'][0]) == (0 if many and mode == L.SMX_PPO_ADAPT else (rows + 15) // 16)
    for k in ('dz3aT', 'dz2aT', 'dz1aT', 'dz2cT', 'dz1cT'):
        close(one[k], two[k], atol=1e-8, rtol=1e-5, msg=k)
    for k in ('grads_a', 'grads_c'):
        close(one[k], two[k], atol=1e-7, rtol=2e-5, msg=k)


@pytest.mark.parametrize('mode', [L.SMX_PPO_ADAPT, L.SMX_PPO_CLIP])
def test_epoch_fwdbwd_hand_off_under_uneven_load_many_times(K, mode):
    """the in-launch publish / poll protocol of smx_epoch_fwdbwd_f32 (device-scope stores of the loss partial rows and the
    KL slots, a counter, device-scope polls) exercised 3000 times at the benchmark shape while a second stream
    keeps the device UNEVENLY busy (tenants of 40 ... 120 CUs for 20 ... 300 us, started at random points): every launch
    must reproduce the statistics, the gradients' right-hand sides and the partial rows of the first one bit for bit, the
    counter must count every actor workgroup, and no wait may time out"""
    rows, D, H1, H2, A = 1024, 376, 300, 200, 17
    t = build(rows, D, H1, H2, A, seed=99, mode=mode, device='cuda')['d']
    t['ctrl'][L.C_KL_TARGET] = 1e9
    nb = (rows + 15) // 16
    loss = dict(mode=mode, rows=rows, log_var=t['log_var'], actions=t['actions'], behave=t['behave'], ref=t['ref'],
                adv=t['adv'], g_surr=t['g_surr'], g_kl=t['g_kl'], partials=t['partials'], check_stop=True,
                will_update=True, dlogvar=t['dlogvar'], dlogvar_sumsq=t['dlq'], stats=t['stats'],
                returns=t['returns'], v_dz3=t['v_dz3'], v_partials=t['v_partials'], v_will_update=True)
    K.epoch_pack([(t['act'], t['pk_a']), (t['cri'], t['pk_c'])])
    aj = dict(net=t['act'], packed=t['pk_a'], x=t['x'], h1T=t['h1aT'], h2T=t['h2aT'], act=L.SMX_ACT_TANH, loss='policy',
              dz3T=t['dz3aT'], dz2T=t['dz2aT'], dz1T=t['dz1aT'], xT=t['xT'], grads=t['grads_a'])
    cj = dict(net=t['cri'], packed=t['pk_c'], x=t['x'], h1T=t['h1cT'], h2T=t['h2cT'], act=L.SMX_ACT_NONE,
              loss='value', dz3=t['v_dz3'], dz3T=t['v_dz3'], dz2T=t['dz2cT'], dz1T=t['dz1cT'], xT=t['xT'],
              grads=t['grads_c'])
    sync = torch.zeros(1, dtype=torch.int32, device='cuda')
    slots = torch.zeros(2 * nb, dtype=torch.int32, device='cuda')
    ctrl0 = t['ctrl'].clone()
    keys = ('stats', 'dlogvar', 'partials', 'v_partials', 'dz3aT', 'dz2aT', 'dz1aT', 'dz2cT', 'dz1cT')
    side = torch.cuda.Stream()
    rs = np.random.RandomState(7)
    first = None
    for it in range(3000):
        sync.zero_(); slots.zero_(); t['ctrl'].copy_(ctrl0)
        for k in ('stats', 'dz1aT', 'dz1cT'):
            t[k].fill_(123.0)             # (a stale result of the previous launch must not pass for a fresh one)
        if rs.rand() < 0.7:
            with torch.cuda.stream(side):
                K.device_occupy(int(rs.randint(40, 121)), int(rs.randint(20, 301)))
        K.epoch_fwdbwd([aj, cj], loss, t['ctrl'], rows, sync, slots)
        if it % 50 == 0 or it < 5:                        # (a device sync every launch would remove the unevenness)
            torch.cuda.synchronize()
            assert int(t['ctrl'].view(torch.int32)[L.C_SYNC_ERR]) == 0, it
            assert int(sync[0]) == nb, (it, int(sync[0]))
            snap = {k: t[k].clone() for k in keys}
            if first is None:
                first = snap
            for k in keys:
                assert torch.equal(snap[k], first[k]), (it, k)
    torch.cuda.synchronize()
    assert int(t['ctrl'].view(torch.int32)[L.C_SYNC_ERR]) == 0


@pytest.mark.parametrize('mode', ['adapt', 'clip'])
@pytest.mark.parametrize('rows,D,H1,H2,A', [(1024, 376, 300, 200, 17), (100, 64, 332, 212, 32), (48, 12, 24, 16, 3)])
def test_epoch_launches_against_the_reference_restatement_in_float64(K, rows, D, H1, H2, A, mode):
    """An oracle the kernels share nothing with: the reference's losses as oracle/ppo_oracle.py restates them
    (_adapt_loss / _clip_loss / _value_loss, surreal/learner/ppo.py:194-332) in float64 with torch autograd, against ONE
    epoch of [smx_epoch_fwdbwd_f32 -> smx_mlp3_wgrad_multi_f32]: every gradient tensor of both networks and log_var, the
    loss statistics, and the norms clip_grad_norm_ would see.  Tolerances: fp32 sums over <= 1024 rows against float64,
    and hidden units whose pre-activation rounds to the other side of zero (a whole column of a weight gradient then
    moves by one row's contribution): 2e-4 of the largest entry of the tensor."""
    import ppo_oracle
    t = build(rows, D, H1, H2, A, seed=rows + A, mode=mode, device='cuda')['d']
    m = L.SMX_PPO_ADAPT if mode == 'adapt' else L.SMX_PPO_CLIP
    na = t['act'].numel
    ga = torch.zeros(na + A, device='cuda')
    np_a = K.mlp3_backward_partials(t['act'])
    sq_a, sq_c = torch.zeros(np_a + 1, device='cuda'), torch.zeros(K.mlp3_backward_partials(t['cri']), device='cuda')
    t['ctrl'][L.C_KL_TARGET] = 0.015
    sync = torch.zeros(4, dtype=torch.int32, device='cuda')
    kl = torch.zeros(2 * ((rows + 15) // 16), dtype=torch.int32, device='cuda')
    loss = dict(mode=m, rows=rows, log_var=t['log_var'], actions=t['actions'], behave=t['behave'], ref=t['ref'],
                adv=t['adv'], g_surr=t['g_surr'], g_kl=t['g_kl'], partials=t['partials'], check_stop=False,
                will_update=True, dlogvar=ga[na:], dlogvar_sumsq=sq_a[np_a:], stats=t['stats'],
                returns=t['returns'], v_dz3=t['v_dz3'], v_partials=t['v_partials'], v_will_update=True)
    K.epoch_pack([(t['act'], t['pk_a']), (t['cri'], t['pk_c'])])
    aj = dict(net=t['act'], packed=t['pk_a'], x=t['x'], h1T=t['h1aT'], h2T=t['h2aT'], act=L.SMX_ACT_TANH, loss='policy',
              dz3T=t['dz3aT'], dz2T=t['dz2aT'], dz1T=t['dz1aT'], xT=t['xT'], grads=ga, sumsq=sq_a)
    cj = dict(net=t['cri'], packed=t['pk_c'], x=t['x'], h1T=t['h1cT'], h2T=t['h2cT'], act=L.SMX_ACT_NONE, loss='value',
              dz3=t['v_dz3'], dz3T=t['v_dz3'], dz2T=t['dz2cT'], dz1T=t['dz1cT'], xT=t['xT'], grads=t['grads_c'], sumsq=sq_c)
    K.epoch_fwdbwd([aj, cj], loss, t['ctrl'], rows, sync[0:1], kl)
    K.mlp3_wgrad_multi([aj, cj])
    torch.cuda.synchronize()

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        params = collections.OrderedDict()
        for nm, net in (('actor', t['act']), ('critic', t['cri'])):
            for k in (1, 2, 3):
                params['%s.fc%d.W' % (nm, k)] = net.views['W%d' % k].cpu().double().numpy()
                params['%s.fc%d.b' % (nm, k)] = net.views['b%d' % k].cpu().double().numpy()
        params['actor.log_var'] = t['log_var'].cpu().double().numpy().reshape(1, A)
        O = ppo_oracle.OraclePPOLearner(params, A, rows, use_z_filter=False, ppo_mode=mode, kl_target=0.015)
        d64 = lambda k: t[k].cpu().double()  # noqa: E731
        obs = {'low_dim': {'s': d64('x')}}
        if mode == 'adapt':
            pl, st = O._adapt_loss(obs, d64('actions'), d64('adv'), d64('behave'), d64('ref'))
        else:
            pl, st = O._clip_loss(obs, d64('actions'), d64('adv'), d64('behave'))
        pl.backward()
        vl, vst = O._value_loss(obs, d64('returns').view(-1, 1))
        vl.backward()
        P = O.model.p
        want_a = torch.cat([P['actor.fc%d.%s' % (k, w)].grad.reshape(-1) for k in (1, 2, 3) for w in ('W', 'b')] +
                           [P['actor.log_var'].grad.reshape(-1)])
        want_c = torch.cat([P['critic.fc%d.%s' % (k, w)].grad.reshape(-1) for k in (1, 2, 3) for w in ('W', 'b')])
    finally:
        torch.set_default_dtype(prev)
    got_a, got_c = ga.cpu().double(), t['grads_c'].cpu().double()

    bad = []

    def near(got, want, what, who):
        o = 0
        for nm, n in what:
            g, w = got[o:o + n], want[o:o + n]
            tol = 2e-4 * float(w.abs().max()) + 1e-9
            if not float((g - w).abs().max()) <= tol:
                bad.append('%s %s: off by %.3g, tolerance %.3g (largest entry %.3g)' % (who, nm, float((g - w).abs().max()), tol,
                                                                                   float(w.abs().max())))
            o += n
    sizes = lambda net, extra: [('W1', H1 * D), ('b1', H1), ('W2', H2 * H1), ('b2', H2), ('W3', net.OUT * H2),  # noqa: E731
                                ('b3', net.OUT)] + extra
    near(got_a, want_a, sizes(t['act'], [('log_var', A)]), 'actor')
    near(got_c, want_c, sizes(t['cri'], []), 'critic')
    assert not bad, (bad, st, t['stats'].cpu()[:4])
    # what clip_grad_norm_ would see
    assert abs(float(torch.sqrt(sq_a.double().sum())) - float(want_a.norm())) <= 2e-4 * float(want_a.norm())
    assert abs(float(torch.sqrt(sq_c.double().sum())) - float(want_c.norm())) <= 2e-4 * float(want_c.norm())
    # the epoch's statistics (mean-reduced sums of 1024 fp32 terms)
    stats = t['stats'].cpu()
    assert abs(float(stats[L.PS_SURR]) - st['_surr_loss']) <= 2e-5 * max(1.0, abs(st['_surr_loss']))
    assert abs(float(stats[L.PS_ENTROPY]) - st['_entropy']) <= 2e-5 * max(1.0, abs(st['_entropy']))
    if mode == 'adapt':
        assert abs(float(stats[L.PS_KL]) - st['_pol_kl']) <= 2e-5 * max(1.0, abs(st['_pol_kl']))
        assert abs(float(stats[L.PS_LOSS]) - st['_kl_loss_adapt']) <= 5e-5 * max(1.0, abs(st['_kl_loss_adapt']))
    else:
        assert abs(float(stats[L.PS_LOSS]) - st['_clip_surr_loss']) <= 2e-5 * max(1.0, abs(st['_clip_surr_loss']))
    sq_err = float(t['v_partials'].cpu().double()[:, 5].sum())          # (word 5 of a block's row: its sum of squared errors)
    assert abs(sq_err / rows - vst['_val_loss']) <= 2e-5 * max(1.0, vst['_val_loss'])


def test_epoch_fwdbwd_kl_cutoff_early_exit_and_repeats(K):
    rows, D, H1, H2, A = 200, 24, 64, 48, 6
    # (a) the KL cutoff active: c_kl = beta + 2 eta (KL - 2 kl_target) carries the batch KL through the counter
    two, one = _fb_pair(K, rows, D, H1, H2, A, L.SMX_PPO_ADAPT, seed=11)
    kl = float(two['stats'][L.PS_KL])
    two, one = _fb_pair(K, rows, D, H1, H2, A, L.SMX_PPO_ADAPT, seed=11, kl_target=kl / 3.0)    # 2 kt < KL < 4 kt
    assert int(one['ctrl'].view(torch.int32)[L.C_STOP]) == 0
    # (c_kl is large here and the two shares cancel: the absolute bound scales with |c_kl| * |W3^T g_kl| ~ 1e-2)
    for k in ('dz3aT', 'dz2aT', 'dz1aT'):
        close(one[k], two[k], atol=3e-7, rtol=1e-5, msg=k)
    assert float(one['dz3aT'].abs().sum()) > 0
    # (b) the early exit: flag raised, counters untouched, no actor gradient leaves the launch in adapt mode
    for mode in (L.SMX_PPO_ADAPT, L.SMX_PPO_CLIP):
        two, one = _fb_pair(K, rows, D, H1, H2, A, mode, seed=12, kl_target=1e-7)
        for t in (one, two):
            ci = t['ctrl'].cpu().view(torch.int32)
            assert int(ci[L.C_STOP]) == 1 and int(ci[L.C_EPOCHS_DONE]) == 0 and int(ci[L.C_STEP_ACTOR]) == 0
            assert int(ci[L.C_STEP_CRITIC]) == 1
        close(one['stats'], two['stats'], atol=1e-6, rtol=2e-6)
        close(one['dz1cT'], two['dz1cT'], atol=1e-8, rtol=1e-5)
        if mode == L.SMX_PPO_ADAPT:
            assert float(one['dz1aT'].abs().sum()) == 0.0
    # (c) three launches in a row on their own counter words give what three pairs of launches give
    two, one = _fb_pair(K, rows, D, H1, H2, A, L.SMX_PPO_ADAPT, seed=13, launches=3)
    assert one['sync'].cpu().tolist() == [(rows + 15) // 16] * 3 + [0]
    assert torch.equal(one['ctrl'].view(torch.int32)[L.C_STEP_ACTOR:], two['ctrl'].view(torch.int32)[L.C_STEP_ACTOR:])
    for k in ('dz2aT', 'dz1aT', 'dz1cT'):
        close(one[k], two[k], atol=1e-8, rtol=1e-5, msg=k)


def test_epoch_early_exit_and_final_pass(K):
    rows, D, H1, H2, A = 64, 16, 40, 24, 4
    # early exit: kl_target so small that KL > 4 kl_target -> flag raised, nothing updated
    T = build(rows, D, H1, H2, A, seed=5, mode=L.SMX_PPO_ADAPT, device='cuda')
    for t in T.values():
        t['ctrl'][L.C_KL_TARGET] = 1e-6
    run(C, T['c'], L.SMX_PPO_ADAPT)
    run(K, T['d'], L.SMX_PPO_ADAPT)
    torch.cuda.synchronize()
    ci, cd = T['c']['ctrl'].view(torch.int32), T['d']['ctrl'].cpu().view(torch.int32)
    assert int(ci[L.C_STOP]) == 1 and int(cd[L.C_STOP]) == 1
    assert int(cd[L.C_EPOCHS_DONE]) == 0 and int(cd[L.C_STEP_ACTOR]) == 0 and int(cd[L.C_STEP_CRITIC]) == 1
    assert float(T['d']['dz1aT'].abs().sum()) == 0.0             # the actor's backward did not run
    close(T['d']['dz1cT'], T['c']['dz1cT'], atol=1e-7, rtol=2e-5)  # the critic's did
    close(T['d']['stats'], T['c']['stats'], atol=1e-5, rtol=2e-5)
    # a raised flag turns the actor job of the next forward into a no-op
    T['d']['mean'].fill_(7.0)
    run(K, T['d'], L.SMX_PPO_ADAPT, stop=T['d']['ctrl'].view(torch.int32)[L.C_STOP:L.C_STOP + 1])
    torch.cuda.synchronize()
    assert float(T['d']['mean'].min()) == 7.0
    # final, forward-only pass: statistics only
    T = build(rows, D, H1, H2, A, seed=6, mode=L.SMX_PPO_CLIP, device='cuda')
    run(C, T['c'], L.SMX_PPO_CLIP, check_stop=True, will_update=False)
    run(K, T['d'], L.SMX_PPO_CLIP, check_stop=True, will_update=False)
    torch.cuda.synchronize()
    close(T['d']['stats'], T['c']['stats'], atol=1e-5, rtol=2e-5)
    assert int(T['d']['ctrl'].cpu().view(torch.int32)[L.C_EPOCHS_DONE]) == 0
    assert float(T['d']['dz1aT'].abs().sum()) == 0.0


@pytest.mark.parametrize('rows,N,D,A,use_z,tail', [(1024, 3, 376, 17, True, True), (37, 2, 29, 5, True, False),
                                                  (16, 1, 12, 3, False, True)])
def test_epoch_prepare_equals_the_separate_launches(K, rows, N, D, A, use_z, tail):
    """smx_epoch_prepare_f32 (one launch) against the z-filter / copy / exp / pack steps it replaces"""
    import types
    g = torch.Generator().manual_seed(rows + D)
    obs = torch.randn(rows, N, D, generator=g) * 2 + 0.3
    obs_next = torch.randn(rows, 1, D, generator=g)
    zmean, zstd = torch.randn(D, generator=g) * 0.2, torch.rand(D, generator=g) + 0.5
    ref = types.SimpleNamespace(running_sum=torch.randn(D, generator=g) * 50, running_sumsq=torch.rand(D, generator=g) * 4000 + 500,
                                count=torch.tensor([1000.0]), eps=1e-5)
    log_var = -1.0 + 0.3 * torch.randn(A, generator=g)
    act_c, act_d = make_net(D, 24, 16, A, 5, 'cuda')
    cri_c, cri_d = make_net(D, 24, 16, 1, 6, 'cuda')

    def side(dv, Kx, act, cri):
        t = lambda x: x.to(dv)  # noqa: E731
        f = lambda *s: torch.full(s, 7.0, device=dv)  # noqa: E731
        o, on = t(obs), t(obs_next)
        out = dict(xn=f(rows, D), xnT=torch.full((D, rows + 16), 7.0, device=dv)[:, :rows], xr=f(rows, D), xnext=f(rows, D),
                   ref_pol=f(rows, 2 * A), zero=torch.full((20,), 3, dtype=torch.int32, device=dv),
                   pa=f(max(Kx.epoch_packed_numel(act), act.numel)), pc=f(max(Kx.epoch_packed_numel(cri), cri.numel)))
        rf = types.SimpleNamespace(running_sum=t(ref.running_sum), running_sumsq=t(ref.running_sumsq), count=t(ref.count),
                                   eps=ref.eps)
        Kx.epoch_prepare(o[:, 0, :], out['xn'], out['xnT'], out['xr'], zmean=t(zmean) if use_z else None,
                         zstd=t(zstd) if use_z else None, ref_filter=rf if use_z else None,
                         obs_next=on[:, 0, :] if tail else None, xnext=out['xnext'] if tail else None,
                         ref_log_var=t(log_var), ref_std=out['ref_pol'][:, A:], pack=[(act, out['pa']), (cri, out['pc'])],
                         zero_words=out['zero'][4:15])
        return out
    c = side('cpu', C, act_c, cri_c)
    d = side('cuda', K, act_d, cri_d)
    torch.cuda.synchronize()
    for k in ('xn', 'xnT', 'xr', 'xnext', 'ref_pol'):
        close(d[k], c[k], atol=1e-6, rtol=1e-6, msg=k)
    # ... and bit for bit what the separate HIP launches give
    o = obs.cuda()
    if use_z:
        ref_d = types.SimpleNamespace(**{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in vars(ref).items()})
        x1, x2 = torch.empty(rows, D).cuda(), torch.empty(rows, D).cuda()
        K.zfilter_forward(o[:, 0, :], zmean.cuda(), zstd.cuda(), x1)
        K.zfilter_forward_sums(o[:, 0, :], ref_d.running_sum, ref_d.running_sumsq, ref_d.count, ref_d.eps, x2)
        assert torch.equal(x1, d['xn']) and torch.equal(x1.t(), d['xnT']) and torch.equal(x2, d['xr'])
        if tail:
            K.zfilter_forward(obs_next.cuda()[:, 0, :], zmean.cuda(), zstd.cuda(), x1)
            assert torch.equal(x1, d['xnext'])
    else:
        assert torch.equal(o[:, 0, :], d['xn']) and torch.equal(o[:, 0, :], d['xr'])
    assert float(d['ref_pol'][:, :A].min()) == 7.0                          # the mean columns are not touched
    assert d['zero'].cpu().tolist() == [3] * 4 + [0] * 11 + [3] * 5
    if not tail:
        assert float(d['xnext'].min()) == 7.0
    # the packed copies: the forward kernel must see the same weights as through smx_epoch_pack_f32
    pa2 = torch.zeros_like(d['pa'])
    K.epoch_pack([(act_d, pa2)])
    assert torch.equal(pa2, d['pa'])


@pytest.mark.parametrize('B,N', [(1024, 128), (37, 19), (5, 3)])
def test_gae_norm_equals_the_three_launches(K, B, N):
    g = torch.Generator().manual_seed(B + N)
    values = torch.randn(B * N, generator=g) * 3
    tail = torch.randn(B, generator=g)
    rewards = torch.randn(B, N, generator=g)
    dones = (torch.rand(B, N, generator=g) < 0.1).float()
    idx = torch.tensor(range(N), dtype=torch.float32)
    gpow, lpow = torch.pow(0.995, idx), torch.pow(0.97, idx)
    ac, rc, mc = torch.empty(B), torch.empty(B), torch.empty(3)
    C.gae_norm(values, rewards, dones, gpow, lpow, 0.995, 0.995 ** N, B, N, N, ac, rc, mc, 1e-4, None, values_tail=tail)
    ad, rd, md = torch.empty(B).cuda(), torch.empty(B).cuda(), torch.empty(3).cuda()
    ticket = torch.zeros(1, dtype=torch.int32).cuda()
    for _ in range(2):                 # twice: the ticket counter must come back to zero
        K.gae_norm(dev(values), dev(rewards), dev(dones), dev(gpow), dev(lpow), 0.995, 0.995 ** N, B, N, N, ad, rd, md,
                   1e-4, ticket, values_tail=dev(tail))
    close(rd, rc, msg='ret'), close(md, mc, atol=1e-5, rtol=1e-5, msg='moments'), close(ad, ac, msg='normalised adv')
    assert int(ticket[0]) == 0


@pytest.mark.parametrize('rows,D,A,Ev,nblk,use_z', [(1024, 376, 17, 10, 64, True), (37, 29, 5, 3, 3, True),
                                                   (64, 17, 6, 20, 4, False)])
def test_learn_epilogue_equals_the_four_launches(K, rows, D, A, Ev, nblk, use_z):
    import types
    g = torch.Generator().manual_seed(rows + Ev)
    x3 = torch.randn(rows, 2, D, generator=g) * 2 + 0.5
    ret = torch.randn(rows, generator=g) * 3 + 1
    part = torch.zeros(Ev, nblk, 8)
    part[:, :, 0] = 16.0
    part[:, :, 1:6] = torch.rand(Ev, nblk, 5, generator=g) + 0.1
    log_var = -1.0 + 0.3 * torch.randn(A, generator=g)
    zf0 = dict(running_sum=torch.randn(D, generator=g) * 100, running_sumsq=torch.rand(D, generator=g) * 5000 + 2000,
               count=torch.tensor([1000.0]), eps=1e-5)

    def side(dv, Kx):
        t = lambda v: v.to(dv).clone() if torch.is_tensor(v) else v  # noqa: E731
        zf = types.SimpleNamespace(**{k: t(v) for k, v in zf0.items()}) if use_z else None
        out = dict(ret_mom=torch.zeros(3, device=dv), vstats=torch.zeros(Ev, L.VS_STRIDE, device=dv),
                   out4=torch.zeros(4, device=dv), zf=zf, ticket=torch.zeros(1, dtype=torch.int32, device=dv))
        Kx.learn_epilogue(t(ret), out['ret_mom'], t(log_var), out['out4'], out['ticket'], zfilter=zf,
                          x=t(x3)[:, 0, :] if use_z else None, count_rows=rows, v_partials=t(part), n_epochs=Ev,
                          nblk=nblk, v_stats=out['vstats'], stats_stride=L.VS_STRIDE)
        return out
    c, d = side('cpu', C), side('cuda', K)
    torch.cuda.synchronize()
    close(d['ret_mom'], c['ret_mom'], atol=1e-5, rtol=1e-5)
    close(d['vstats'][:, :2], c['vstats'][:, :2], atol=1e-5, rtol=1e-5)
    close(d['out4'], c['out4'], atol=1e-5, rtol=2e-5, msg='reported means (after the z-filter update)')
    if use_z:
        close(d['zf'].running_sum, c['zf'].running_sum, rtol=1e-5, atol=1e-3)
        close(d['zf'].running_sumsq, c['zf'].running_sumsq, rtol=1e-5, atol=1e-2)
        close(d['zf'].count, c['zf'].count)
    assert int(d['ticket'][0]) == 0
