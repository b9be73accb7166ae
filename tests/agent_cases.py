"""Shared body of the agent parity tests (tests only): the product's rollout workers against
tests/golden/agents.npz, recorded from the REFERENCE's own PPOAgent / DDPGAgent / action-noise
classes (oracle/gen_golden_agents.py -- real constructors, real ``act``).  The CPU tier runs it
on the torch-CPU kernel double, the GPU tier (-m gpu) on the HIP kernels through the C ABI.

Checked per case: the constructor's mode remap and exploration-noise draw, every step's action
and policy distribution ``[mean | std * exp(noise)]``, the LSTM state handed to the windowing
wrapper (``onetime_infos``) incl. across ``reset()``, the ``action_info`` layout, and the batched
device path ``act_batch`` (all actors in one launch chain) with the reference's normal draws
injected.  Tolerance 1e-5 (fp32, BASELINE.json's bound) on pd / actions."""
import collections
import contextlib
import io
import json
import os

import numpy as np
import torch

import helpers as H
from surreal_amd import synthetic
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config

import ddpg_oracle

G = np.load(os.path.join(H.GOLDEN_DIR, 'agents.npz'))
META = json.loads(str(G['meta_json']))
PPO_CASES = list(META['ppo'])
DDPG_CASES = list(META['ddpg'])
ATOL = RTOL = 1e-5


def _ppo_setup(c):
    lc = ppo_learner_config()
    lc.model.actor_fc_hidden_sizes = list(c['hidden'])
    lc.model.critic_fc_hidden_sizes = list(c['hidden'])
    lc.model.cnn_feature_dim = c.get('cnn_feature_dim', 256)
    lc.algo.use_z_filter = c.get('use_z', True)
    rnn = c.get('rnn')
    lc.algo.rnn.if_rnn_policy = rnn is not None
    if rnn:
        lc.algo.rnn.rnn_hidden, lc.algo.rnn.rnn_layer = rnn
    pixel = tuple(c['pixel']) if c.get('pixel') else None
    ec = ppo_env_config(c['D'], c['A'], pixel=pixel)
    ec.stochastic_eval = c.get('stochastic_eval', True)
    pix_kw = dict(pixel=pixel, cnn_feature_dim=c['cnn_feature_dim']) if pixel else {}
    params = synthetic.make_ppo_params(c['D'], c['A'], hidden=tuple(c['hidden']),
                                       rnn_hidden=rnn[0] if rnn else 0, rnn_layers=rnn[1] if rnn else 1,
                                       **META['ppo_param_kw'], **pix_kw)
    zstate = synthetic.make_zfilter_state(c['D'], seed=5) if c.get('use_z', True) else None
    return lc, ec, ppo_session_config('/tmp/surreal_amd_test'), params, zstate


def _obs(c, g, t, i):
    o = collections.OrderedDict()
    if c.get('pixel'):
        o['pixel'] = collections.OrderedDict(camera0=g('obs_pix')[t, i])
    o['low_dim'] = collections.OrderedDict(flat_inputs=g('obs_low')[t, i])
    return o


def check_ppo_case(name):
    from surreal_amd.agent import PPOAgent
    c = META['ppo'][name]
    g = lambda k: G['ppo.%s.%s' % (name, k)]  # noqa: E731
    lc, ec, sc, params, zstate = _ppo_setup(c)
    A, n = c['A'], c['n_agents']
    T = c['steps'] + c.get('steps2', 0)
    rnn = c.get('rnn')
    agents = []
    for i in range(n):
        np.random.seed(100 + i)
        ag = PPOAgent(lc, ec, sc, agent_id=i, agent_mode=c['mode'])
        assert ag.agent_mode == c['resolved_modes'][i]                       # ppo_agent.py:49-55
        assert ag.noise == g('noise')[i]                                     # :57-61, bit-exact draw
        ag.model.load_params(params)
        if zstate is not None:
            ag.model.z_filter.load_state_dict(zstate)
        np.random.seed(300 + i)
        for t in range(T):
            if c.get('reset_after') and t == c['reset_after']:
                ag.reset()
            got = ag.act(_obs(c, g, t, i))
            if ag.agent_mode == 'training':
                a, info = got
                assert isinstance(info, list) and len(info) == 2 and len(info[1]) == 1
                pd = info[1][0]
                assert pd.shape == (2 * A,) and pd.dtype == np.float32
                np.testing.assert_allclose(pd, g('pds')[t, i], atol=ATOL, rtol=RTOL,
                                           err_msg='%s agent %d step %d pd' % (name, i, t))
                if rnn:
                    assert len(info[0]) == 2
                    for j in (0, 1):
                        assert info[0][j].shape == (rnn[1], rnn[0]) and info[0][j].dtype == np.float32
                        np.testing.assert_allclose(info[0][j], g('cells_before')[t, i, j], atol=ATOL, rtol=RTOL,
                                                   err_msg='%s agent %d step %d cell %d' % (name, i, t, j))
                else:
                    assert info[0] == []
            else:
                a = got
            assert isinstance(a, np.ndarray) and a.shape == (A,) and str(a.dtype) == c['action_dtype']
            np.testing.assert_allclose(a, g('actions')[t, i], atol=ATOL, rtol=RTOL,
                                       err_msg='%s agent %d step %d action' % (name, i, t))
        agents.append(ag)
    # ---- the same rollout, all n actors per step through act_batch, reference draws injected -----
    ag = agents[0]
    dev = ag.device
    ag.reset_batch()
    ag._batch_noise = torch.as_tensor(np.exp(g('noise')), dtype=torch.float32).view(n, 1).to(dev)
    for t in range(T):
        if c.get('reset_after') and t == c['reset_after']:
            ag.reset_batch(torch.ones(n, dtype=torch.bool, device=dev))
        low = torch.as_tensor(g('obs_low')[t]).to(dev)
        obs = low
        if c.get('pixel'):
            obs = {'pixel': {'camera0': torch.as_tensor(g('obs_pix')[t]).to(dev)},
                   'low_dim': {'flat_inputs': low}}
        eps = torch.as_tensor(g('eps')[t], dtype=torch.float32).to(dev)
        acts, pds = ag.act_batch(obs, eps=eps)
        if c['resolved_modes'][0] == 'training':
            np.testing.assert_allclose(pds.cpu().numpy(), g('pds')[t], atol=ATOL, rtol=RTOL,
                                       err_msg='%s act_batch step %d pd' % (name, t))
        # eps was drawn in fp64 by the reference and is injected as fp32: 1e-7 relative on std * eps
        np.testing.assert_allclose(acts.cpu().numpy(), g('actions')[t], atol=ATOL, rtol=RTOL,
                                   err_msg='%s act_batch step %d action' % (name, t))
        if rnn:
            hb, cb = ag.batch_cells_before
            np.testing.assert_allclose(hb.permute(1, 0, 2).cpu().numpy(), g('cells_before')[t, :, 0],
                                       atol=ATOL, rtol=RTOL)
            np.testing.assert_allclose(cb.permute(1, 0, 2).cpu().numpy(), g('cells_before')[t, :, 1],
                                       atol=ATOL, rtol=RTOL)


def _ddpg_setup(c):
    lc = ddpg_learner_config()
    lc.model.actor_fc_hidden_sizes = list(c['ah'])
    lc.model.critic_fc_hidden_sizes = list(c['ch'])
    lc.model.conv_spec.hidden_output_dim = c.get('conv_hidden', 200)
    lc.algo.exploration.noise_type = c['noise_type']
    pixel = tuple(c['pixel']) if c.get('pixel') else None
    ec = ddpg_env_config(c['D'], c['A'], num_agents=c['num_agents'], pixel=pixel)
    ec.frame_stack_concatenate_on_env = not c.get('frame_list')
    if pixel:
        params = ddpg_oracle.make_ddpg_pixel_params(c['D'], c['A'], pixel, c['conv_hidden'], tuple(c['ah']),
                                                    tuple(c['ch']), seed=3)
    else:
        params = ddpg_oracle.make_ddpg_params(c['D'], c['A'], tuple(c['ah']), tuple(c['ch']), seed=3)
    return lc, ec, ddpg_session_config(), params


def check_ddpg_case(name):
    from surreal_amd.agent import DDPGAgent
    c = META['ddpg'][name]
    g = lambda k: G['ddpg.%s.%s' % (name, k)]  # noqa: E731
    lc, ec, sc, params = _ddpg_setup(c)
    np.random.seed(17)
    with contextlib.redirect_stdout(io.StringIO()):
        ag = DDPGAgent(lc, ec, sc, agent_id=c['agent_id'], agent_mode=c['mode'])
    assert ag.agent_mode == c['resolved_mode'] and ag.sigma == c['sigma']         # ddpg_agent.py:78-84
    ag.model.load_params(params)
    stochastic = ag.agent_mode not in ('eval_deterministic', 'eval_deterministic_local')
    np.random.seed(500)
    t = 0
    fl = c.get('frame_list', 0)
    for ep_len in c['episodes']:
        ag.pre_episode()                                                          # :205-208 (OU state reset)
        for _ in range(ep_len):
            o = collections.OrderedDict()
            if c.get('pixel'):
                fr = g('obs_pix')[t]
                if fl:
                    k = fr.shape[0] // fl
                    fr = [fr[i * k:(i + 1) * k] for i in range(fl)]
                o['pixel'] = collections.OrderedDict(camera0=fr)
            o['low_dim'] = collections.OrderedDict(flat_inputs=g('obs_low')[t])
            a = ag.act(o)
            assert isinstance(a, np.ndarray) and a.shape == (c['A'],) and str(a.dtype) == c['action_dtype']
            np.testing.assert_allclose(a, g('actions')[t], atol=ATOL, rtol=RTOL,
                                       err_msg='%s step %d' % (name, t))
            t += 1
    # batched device path with the reference's draws injected (gaussian exploration only: the OU
    # process is stateful per actor and stays on the host)
    if c['noise_type'] == 'normal' and not c.get('pixel'):
        obs = torch.as_tensor(g('obs_low')).to(ag.device)
        eps = torch.as_tensor(g('eps'), dtype=torch.float32).to(ag.device)
        acts = ag.act_batch(obs, eps=eps if stochastic else None)
        np.testing.assert_allclose(acts.cpu().numpy(), g('actions'), atol=ATOL, rtol=RTOL,
                                   err_msg=name + ' act_batch')


def check_noise_streams():
    from surreal_amd.agent import action_noise as AN
    np.random.seed(31)
    nn = AN.NormalActionNoise(np.array([0.1, -0.2, 0.0]), np.array([0.5, 1.0, 2.0]))
    np.testing.assert_array_equal(np.stack([nn() for _ in range(6)]), G['noise.normal'])
    assert repr(nn) == META['noise_repr']['normal']
    np.random.seed(32)
    ou = AN.OrnsteinUhlenbeckActionNoise(mu=np.array([0.2, -0.1]), sigma=0.3, theta=0.15, dt=1e-2)
    a = [ou() for _ in range(5)]
    ou.reset()
    a += [ou() for _ in range(3)]
    np.testing.assert_array_equal(np.stack(a), G['noise.ou'])
    np.random.seed(33)
    ou0 = AN.OrnsteinUhlenbeckActionNoise(mu=np.zeros(2), sigma=np.array([0.3, 0.6]), theta=0.5, dt=0.25,
                                          x0=np.array([1.0, -1.0]))
    np.testing.assert_array_equal(np.stack([ou0() for _ in range(4)]), G['noise.ou_x0'])
    assert repr(ou0) == META['noise_repr']['ou']
