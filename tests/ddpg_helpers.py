import json
import os

import numpy as np

import helpers as H
from surreal_amd import synthetic
from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config

import ddpg_oracle

DDPG_CASES = ['tiny_hard', 'tiny_soft_clipcritic', 'tiny_td3_hard', 'tiny_double_soft', 'tiny_pixel_hard',
              'tiny_pixel_td3_soft', 'cfg3_cheetah512', 'tiny_ln_hard', 'ln_soft_clipcritic', 'cfg3_cheetah512_x502',
              'tiny_ln_td3_soft', 'tiny_ln_pixel_hard', 'tiny_ln_pixel_td3_soft']


def load(name):
    g = np.load(os.path.join(H.GOLDEN_DIR, 'ddpg_%s.npz' % name))
    return g, json.loads(str(g['case_json']))


def make_learner(case, opts=None):
    from surreal_amd.learner.ddpg import DDPGLearner
    h = case['hyper']
    lc = ddpg_learner_config()
    lc.model.actor_fc_hidden_sizes = list(case['ah'])
    lc.model.critic_fc_hidden_sizes = list(case['ch'])
    lc.algo.gamma, lc.algo.n_step = h['gamma'], h['n_step']
    lc.algo.network.lr_actor, lc.algo.network.lr_critic = h['lr_actor'], h['lr_critic']
    lc.algo.network.clip_critic_gradient = h.get('clip_critic', False)
    lc.algo.network.target_update = {'type': h['target_update_type'],
                                     'interval': h['target_update_interval'], 'tau': h.get('tau', 1e-3)}
    lc.algo.network.use_double_critic = bool(h.get('double_critic', False))
    lc.algo.network.use_action_regularization = bool(h.get('action_reg', False))
    lc.model.use_layernorm = bool(h.get('layernorm', False))
    lc.replay.batch_size = case['B']
    pixel = tuple(case['pixel']) if case.get('pixel') else None
    if pixel is not None:
        lc.model.conv_spec.hidden_output_dim = case['conv_hidden']
    sc = ddpg_session_config()
    for k, v in (opts or {}).items():          # e.g. ddpg_row_schedule = False: the level schedule
        sc.learner[k] = v
    L = DDPGLearner(lc, ddpg_env_config(case['D'], case['A'], pixel=pixel), sc)

    def mkp(seed):
        if pixel is not None:
            return ddpg_oracle.make_ddpg_pixel_params(case['D'], case['A'], pixel, case['conv_hidden'],
                                                      tuple(case['ah']), tuple(case['ch']), seed=seed,
                                                      layernorm=bool(h.get('layernorm', False)))
        return ddpg_oracle.make_ddpg_params(case['D'], case['A'], tuple(case['ah']), tuple(case['ch']), seed=seed,
                                            layernorm=bool(h.get('layernorm', False)))
    params = mkp(3)
    L.model.load_params(params)
    L.model_target.load_params(params)
    if L.use_double_critic:
        params2 = mkp(4)
        L.model2.load_params(params2)
        L.model_target2.load_params(params2)
    return L


# Parameters after the golden's iterations: EVERY element of every tensor (model, target, second critic) within
# PARAM_ATOL of the reference's.  Round 5 allowed 2 * lr * iters on 3 % of the elements; that allowance is gone.  Where a
# case needs more than 1e-5 the bound below is the measured one and says why.  DDPG_PARAM_REPORT records what was measured
# (tests/conftest.py writes it to gpurun_out/ddpg_params_report_<tier>.json).
PARAM_ATOL_DEFAULT = 1e-5
PARAM_ATOL = {
    # 502 iterations of Adam at lr_critic 1e-3 on fresh batches, run FROM ITERATION 0: a rounding difference in one gradient
    # enters Adam's m / sqrt(v), is carried rather than damped, and flips ReLU masks downstream -- two fp32 evaluations of the
    # same mathematics separate.  Measured with the CPU kernel double, whose ONLY difference from the reference is the
    # summation order of its GEMMs: parameters up to 5.4e-2 apart (critic.fc2.W), statistics up to 2.9e-2 (Q_policy) by
    # iteration 502, 2e-3 by iteration 212.  So this run is a gross-error cap (3 x that) plus the bit-exact properties of
    # the update checked in run_and_check; the 1e-5 statement about the update at this size is
    # check_resume_across_hard_update below (state of iteration 497 loaded, 5 iterations, every element 1e-5).
    'cfg3_cheetah512_x502': 0.16,
    # lr_critic = 1e-2 on these tiny pixel cases: ONE of conv2.W's 8192 elements (1 of fc.W's 3072) has a gradient at the
    # noise floor of its 3.4 k-term sum and takes one Adam step the other way -- on the HIP path AND on the CPU kernel double,
    # by the same 1.43e-4 / 7.6e-5 (gpurun_out/ddpg_params_report_{gpu,cpu}.json); every other element of every tensor <= 1e-5
    'tiny_pixel_hard': 3e-4, 'tiny_pixel_td3_soft': 2e-4,
}
# statistics of the long case: the first iterations at the common 1e-5, later ones at the drift's scale (as above)
LATE_STATS = {'cfg3_cheetah512_x502': dict(after=10, atol=0.09, rtol=0.0)}
DDPG_PARAM_REPORT = {}


def _assert_params(name, which, got, g, prefix):
    atol = PARAM_ATOL.get(name.split(' ')[0], PARAM_ATOL_DEFAULT)
    for k in got:
        if prefix + k not in g:
            continue
        d = np.abs(got[k] - g[prefix + k])
        DDPG_PARAM_REPORT['%s %s %s' % (name, which, k)] = (float(d.max()), float(np.mean(d > 1e-5)), int(d.size), atol)
        assert H.MEASURE_ONLY or d.max() <= atol, '%s %s %s: max diff %g (bound %g), %.3f%% of elements off by > 1e-5' % (
            name, which, k, d.max(), atol, 100 * np.mean(d > 1e-5))


def run_and_check(name, atol=1e-5, rtol=1e-5, opts=None):
    import torch
    g, case = load(name)
    L = make_learner(case, opts)
    ref = json.loads(str(g['trace_json']))
    late = LATE_STATS.get(name)
    interval = case['hyper']['target_update_interval'] if case['hyper']['target_update_type'] == 'hard' else None
    at_update = None
    worst = {}
    for it in range(case['iters']):
        b = synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it,
                                      pixel=tuple(case['pixel']) if case.get('pixel') else None)
        np.random.seed(1000 + it)          # TD3's action-regularisation noise (numpy's global stream)
        st = L.learn(b)
        assert set(st) == set(ref[it]), (sorted(st), sorted(ref[it]))
        a_, r_ = (late['atol'], late['rtol']) if late and it >= late['after'] else (atol, rtol)
        for k, v in ref[it].items():
            worst[k] = max(worst.get(k, 0.0), abs(float(st[k]) - v))
            if H.MEASURE_ONLY and late and it >= late['after']:
                continue
            np.testing.assert_allclose(st[k], v, atol=a_, rtol=r_,
                                       err_msg='%s iteration %d %s' % (name, it, k))
        if interval and case['iters'] > interval and (it + 1) % interval == 0:
            # the model as the hard update (ddpg.py:418-428) must have copied it
            at_update = (L.model.actor_flat.clone(), L.model.critic_flat.clone())
    if late:
        DDPG_PARAM_REPORT['%s statistics worst abs diff' % name] = worst
    got = L.model.numpy_params()
    ss = json.loads(str(g['final_sumsq_json']))
    for k, v in ss.items():
        np.testing.assert_allclose(np.sum(got[k].astype(np.float64) ** 2), v,
                                   rtol=max(2e-4, 20 * PARAM_ATOL.get(name, 0.0)), err_msg=k)
    _assert_params(name, 'model', got, g, 'final.')
    _assert_params(name, 'target', L.model_target.numpy_params(), g, 'target.')
    if L.use_double_critic:
        _assert_params(name, 'model2', L.model2.numpy_params(), g, 'final2.')
        _assert_params(name, 'target2', L.model_target2.numpy_params(), g, 'target2.')
    if at_update is not None:
        # crossed the interval at the case's real size: the target IS the model of that iteration, bit for bit, and the
        # model has moved on since (iters is not a multiple of the interval)
        assert torch.equal(L.model_target.actor_flat, at_update[0]) and torch.equal(L.model_target.critic_flat, at_update[1])
        if case['iters'] % interval:
            assert not torch.equal(L.model.critic_flat, at_update[1])
    return L


def make_oracle(case):
    h = case['hyper']
    mk = lambda seed: ddpg_oracle.make_ddpg_params(case['D'], case['A'], tuple(case['ah']), tuple(case['ch']), seed=seed,  # noqa: E731
                                                   layernorm=bool(h.get('layernorm', False)))
    return ddpg_oracle.OracleDDPGLearner(
        mk(3), gamma=h['gamma'], n_step=h['n_step'], lr_actor=h['lr_actor'], lr_critic=h['lr_critic'],
        clip_critic_gradient=h.get('clip_critic', False), target_update_type=h['target_update_type'],
        target_update_interval=h['target_update_interval'], tau=h.get('tau', 1e-3), batch_size=case['B'])


def load_oracle_state(L, O, steps):
    """put the product learner where the oracle is after `steps` iterations -- parameters, target, both optimisers' Adam
    moments and step counts, the update counter: what resuming a checkpoint of that iteration restores (ddpg.py:383-387
    plus the optimiser state torch.save keeps)"""
    import torch
    L.model.load_params(O.model.numpy_params())
    L.model_target.load_params(O.model_target.numpy_params())
    named = L.model.named_parameters()
    for optim, flat, m, v in ((O.actor_optim, L.model.actor_flat, L.actor_exp_avg, L.actor_exp_avg_sq),
                              (O.critic_optim, L.model.critic_flat, L.critic_exp_avg, L.critic_exp_avg_sq)):
        by_id = {id(p): st for p, st in optim.state.items()}
        for k, view in named.items():
            st = by_id.get(id(O.model.p[k]))
            if st is None:
                continue
            off = (view.data_ptr() - flat.data_ptr()) // 4
            if not (0 <= off and off + view.numel() <= flat.numel()):
                continue
            m[off:off + view.numel()].copy_(st['exp_avg'].reshape(-1))
            v[off:off + view.numel()].copy_(st['exp_avg_sq'].reshape(-1))
            assert int(st['step']) == steps
    L.critic_step = L.actor_step = L.current_iteration = steps
    L.target_update_counter = steps
    torch.cuda.synchronize() if torch.cuda.is_available() else None


def check_resume_across_hard_update(name='cfg3_cheetah512_x502', start=497):
    """configs[2]'s size across the hard target update at the reference's interval, held TIGHT: the restatement (bit-identical
    to the reference where the golden was recorded) is run to iteration `start` on this host, its state is loaded into the
    product learner like a checkpoint, and both take the remaining iterations on the same batches -- statistics at 1e-5,
    every parameter and every target element at 1e-5, the target being the model of iteration 500 and not 502's.  (Run
    from iteration 0 the two fp32 paths drift apart by ~1e-3 over 500 Adam steps, as two x86 hosts running the reference
    do: run_and_check holds that run to the drift's scale, this one holds the update itself to the parity bar.)"""
    import torch
    g, case = load(name)
    ref = json.loads(str(g['trace_json']))
    O = make_oracle(case)
    drift = 0.0
    for it in range(start):
        so = O.learn(synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it))
        drift = max(drift, max(abs(so[k] - ref[it][k]) for k in so))
    DDPG_PARAM_REPORT['%s oracle on this host vs the golden over %d iterations, worst statistic' % (name, start)] = drift
    L = make_learner(case)
    load_oracle_state(L, O, start)
    interval = case['hyper']['target_update_interval']
    snap = None
    for it in range(start, case['iters']):
        b = synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it)
        so, sl = O.learn(b), dict(L.learn(b))
        for k, v in so.items():
            np.testing.assert_allclose(sl[k], v, atol=1e-5, rtol=1e-5, err_msg='%s iteration %d %s' % (name, it, k))
        if (it + 1) % interval == 0:
            snap = (L.model.actor_flat.clone(), L.model.critic_flat.clone())
    assert snap is not None and L.target_update_counter == case['iters']
    assert torch.equal(L.model_target.actor_flat, snap[0]) and torch.equal(L.model_target.critic_flat, snap[1])
    assert not torch.equal(L.model.critic_flat, snap[1])
    for which, got, want in (('model', L.model.numpy_params(), O.model.numpy_params()),
                             ('target', L.model_target.numpy_params(), O.model_target.numpy_params())):
        for k in want:
            d = np.abs(got[k] - want[k])
            DDPG_PARAM_REPORT['%s resumed@%d %s %s' % (name, start, which, k)] = (float(d.max()), float(np.mean(d > 1e-5)), int(d.size), 1e-5)
            assert H.MEASURE_ONLY or d.max() <= 1e-5, (which, k, float(d.max()))
    return L


def check_sampling_into_staging(device):
    """replay.sample_batch(B, out=learner.staging_fields(B)): the sample lands where the captured iteration reads it, no
    copies -- same rows (same Philox counters), same statistics and parameters, bit for bit, as sampling into fresh
    tensors and letting learn() stage them"""
    import torch
    from surreal_amd.replay import UniformReplay
    g, case = load('tiny_hard')
    B, D, A = case['B'], case['D'], case['A']
    learners = [make_learner(case), make_learner(case)]
    lc = ddpg_learner_config()
    lc.replay.batch_size = B
    lc.replay.memory_size = 5000
    gen = torch.Generator().manual_seed(3)
    n = 3000
    fields = {'obs': torch.randn(n, D, generator=gen), 'obs_next': torch.randn(n, D, generator=gen),
              'actions': torch.rand(n, A, generator=gen) * 2 - 1, 'rewards': torch.randn(n, generator=gen),
              'dones': (torch.rand(n, generator=gen) < 0.05).float()}
    replays = []
    for _ in range(2):
        R = UniformReplay(lc, ddpg_env_config(D, A), ddpg_session_config())
        R.insert_batch({k: v.to(device) for k, v in fields.items()})
        replays.append(R)

    def batch(f):
        return {'obs': {'low_dim': {'flat_inputs': f['obs']}}, 'obs_next': {'low_dim': {'flat_inputs': f['obs_next']}},
                'actions': f['actions'], 'rewards': f['rewards'].view(B, 1), 'dones': f['dones'].view(B, 1)}
    for it in range(5):
        f0 = replays[0].sample_batch(B)
        st0 = dict(learners[0].learn(batch(f0)))
        stage = learners[1].staging_fields(B)
        f1 = replays[1].sample_batch(B, out=stage)
        for k in stage:
            assert f1[k].data_ptr() == stage[k].data_ptr(), k
            assert torch.equal(f1[k].reshape(-1), f0[k].reshape(-1)), k
        st1 = dict(learners[1].learn(batch(f1)))
        assert st0 == st1, (it, st0, st1)
    for a, b in ((learners[0].model, learners[1].model), (learners[0].model_target, learners[1].model_target)):
        assert torch.equal(a.actor_flat, b.actor_flat) and torch.equal(a.critic_flat, b.critic_flat)
