import json
import os

import numpy as np

import helpers as H
from surreal_amd import synthetic
from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config

import ddpg_oracle

DDPG_CASES = ['tiny_hard', 'tiny_soft_clipcritic', 'tiny_td3_hard', 'tiny_double_soft', 'tiny_pixel_hard',
              'tiny_pixel_td3_soft', 'cfg3_cheetah512', 'tiny_ln_hard', 'ln_soft_clipcritic']


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
                                                      tuple(case['ah']), tuple(case['ch']), seed=seed)
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


def run_and_check(name, atol=1e-5, rtol=1e-5, opts=None):
    g, case = load(name)
    L = make_learner(case, opts)
    ref = json.loads(str(g['trace_json']))
    for it in range(case['iters']):
        b = synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it,
                                      pixel=tuple(case['pixel']) if case.get('pixel') else None)
        np.random.seed(1000 + it)          # TD3's action-regularisation noise (numpy's global stream)
        st = L.learn(b)
        assert set(st) == set(ref[it]), (sorted(st), sorted(ref[it]))
        for k, v in ref[it].items():
            np.testing.assert_allclose(st[k], v, atol=atol, rtol=rtol,
                                       err_msg='%s iteration %d %s' % (name, it, k))
    got = L.model.numpy_params()
    ss = json.loads(str(g['final_sumsq_json']))
    for k, v in ss.items():
        np.testing.assert_allclose(np.sum(got[k].astype(np.float64) ** 2), v, rtol=2e-4, err_msg=k)
        if 'final.' + k in g:
            d = np.abs(got[k] - g['final.' + k])
            lr = max(case['hyper']['lr_actor'], case['hyper']['lr_critic'])
            assert d.max() <= 2 * lr * case['iters'] + 1e-6, (k, d.max())
            assert np.mean(d > 2e-5) < 0.03, (k, np.mean(d > 2e-5))
    tgt = L.model_target.numpy_params()
    for k in tgt:
        if 'target.' + k in g:
            d = np.abs(tgt[k] - g['target.' + k])
            assert np.mean(d > 2e-5) < 0.03, ('target ' + k, d.max())
    if L.use_double_critic:
        got2, tgt2 = L.model2.numpy_params(), L.model_target2.numpy_params()
        for k in got2:
            np.testing.assert_allclose(got2[k], g['final2.' + k], atol=2 * lr * case['iters'] + 1e-6, err_msg=k)
            assert np.mean(np.abs(got2[k] - g['final2.' + k]) > 2e-5) < 0.03, k
            np.testing.assert_allclose(tgt2[k], g['target2.' + k], atol=2 * lr * case['iters'] + 1e-6, err_msg=k)
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
