"""GPU tier (-m gpu): the fused row-block epoch kernels (csrc/smx_epoch.hip: forward + loss,
finalize + data gradients; smx_mlp3_wgrad_multi_f32) through the C ABI against the torch-CPU statement of
the same contract (tests/cpu_kernels.py) on identical seeded inputs -- shapes of the benchmark
(cfg 5), cfg 2, ragged row counts (partial last row block), widths that are not tile multiples, both
PPO modes, the KL early exit and the forward-only final pass.  1e-5 abs + rel (fp32)."""
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
