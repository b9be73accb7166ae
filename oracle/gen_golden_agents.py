#!/usr/bin/env python
"""
TEST INFRASTRUCTURE ONLY.  tests/golden/agents.npz from the REFERENCE's own rollout workers
(run under oracle/ref_shims.py, build container only):

  * surreal/agent/ppo_agent.py: the real ``PPOAgent.__init__`` (mode remap :49-55, the per-agent
    ``noise ~ U(-log_sig_range, log_sig_range)`` draw :57-61, LSTM cell allocation :83-95) and
    ``PPOAgent.act`` (:106-154) step by step over seeded observations, ``reset`` (:167-183):
    MLP, LSTM (1 and 2 layers), CNN + LSTM policies, every agent_mode, with and without z-filter
  * surreal/agent/ddpg_agent.py: the real ``DDPGAgent.__init__`` (sigma schedule :78-84, noise
    construction :115-147), ``act`` (:155-184) and ``pre_episode`` (:205-208) with gaussian and
    Ornstein-Uhlenbeck exploration, list-of-frames pixel observations
  * surreal/agent/action_noise.py:9-39: raw NormalActionNoise / OrnsteinUhlenbeckActionNoise streams

Parameters are injected (surreal_amd.synthetic / oracle.ddpg_oracle generators, regenerated from
seeds by the tests); numpy's global stream is seeded before construction and before acting, and
the standard-normal draws every ``act`` consumed are recorded (``eps``) so the batched device
path (``act_batch``: all actors of a GPU in one launch chain) can be checked with injected noise.
"""
import collections
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
for _k in ('SYMPH_PS_FRONTEND_HOST', 'SYMPH_PS_FRONTEND_PORT', 'SYMPH_LOGGERPLEX_HOST',
           'SYMPH_LOGGERPLEX_PORT', 'SYMPH_TENSORPLEX_HOST', 'SYMPH_TENSORPLEX_PORT'):
    os.environ.setdefault(_k, '1')          # read by Agent.__init__ / _initialize; nothing connects
from surreal_amd import synthetic  # noqa: E402
import ddpg_oracle  # noqa: E402
import gen_golden as GP  # noqa: E402  (inject_params for the reference PPOModel)
import gen_golden_ddpg as GD  # noqa: E402  (inject for the reference DDPGModel)
from surreal.session import Config  # noqa: E402
from surreal.main.ppo_configs import (PPO_DEFAULT_LEARNER_CONFIG, PPO_DEFAULT_ENV_CONFIG,  # noqa: E402
                                      PPO_DEFAULT_SESSION_CONFIG)
from surreal.main.ddpg_configs import (DDPG_DEFAULT_LEARNER_CONFIG, DDPG_DEFAULT_ENV_CONFIG,  # noqa: E402
                                       DDPG_DEFAULT_SESSION_CONFIG)
from surreal.agent.ppo_agent import PPOAgent  # noqa: E402
from surreal.agent.ddpg_agent import DDPGAgent  # noqa: E402
import surreal.agent.action_noise as AN  # noqa: E402

PPO_CASES = collections.OrderedDict([
    # name: D, A, hidden, mode given to the ctor, stochastic_eval, z-filter, rnn (hidden, layers), pixel
    ('mlp_training', dict(D=11, A=3, hidden=[32, 24], mode='training', n_agents=4, steps=5)),
    ('mlp_noz_training', dict(D=11, A=3, hidden=[32, 24], mode='training', use_z=False, n_agents=2, steps=3)),
    ('mlp_eval_det_local', dict(D=11, A=3, hidden=[32, 24], mode='eval_deterministic_local', n_agents=1, steps=3)),
    ('mlp_eval_stoch_local', dict(D=11, A=3, hidden=[32, 24], mode='eval_stochastic_local', n_agents=1, steps=3)),
    # a non-local eval mode is REMAPPED by env_config.stochastic_eval (ppo_agent.py:49-55)
    ('mlp_eval_remap_stoch', dict(D=11, A=3, hidden=[32, 24], mode='eval_deterministic', stochastic_eval=True,
                                  n_agents=1, steps=2)),
    ('mlp_eval_remap_det', dict(D=11, A=3, hidden=[32, 24], mode='eval_stochastic', stochastic_eval=False,
                                n_agents=1, steps=2)),
    ('lstm_training', dict(D=7, A=2, hidden=[16, 12], mode='training', rnn=(12, 1), n_agents=3, steps=4,
                           reset_after=4, steps2=3)),
    ('lstm2_training', dict(D=7, A=2, hidden=[16, 12], mode='training', rnn=(8, 2), n_agents=2, steps=3,
                            reset_after=3, steps2=2)),
    ('pixel_lstm_training', dict(D=4, A=2, hidden=[16, 12], mode='training', rnn=(12, 1), pixel=(2, 20, 24),
                                 cnn_feature_dim=8, n_agents=2, steps=3)),
    ('cfg5_mlp_training', dict(D=376, A=17, hidden=[300, 200], mode='training', n_agents=3, steps=2)),
])

# injected policy parameters: an output layer large enough that the tanh means are O(0.5), and a
# different log-sigma per action dimension (tests regenerate them with the same arguments)
PPO_PARAM_KW = dict(seed=21, final_scale=3.0, log_sig_spread=0.5)

DDPG_CASES = collections.OrderedDict([
    ('normal_id2of4', dict(D=9, A=3, ah=[16, 12], ch=[20, 16], mode='training', agent_id=2, num_agents=4,
                           noise_type='normal', episodes=[3, 2])),
    ('normal_single', dict(D=9, A=3, ah=[16, 12], ch=[20, 16], mode='training', agent_id=0, num_agents=1,
                           noise_type='normal', episodes=[3])),
    ('ou_id3of4', dict(D=9, A=3, ah=[16, 12], ch=[20, 16], mode='training', agent_id=3, num_agents=4,
                       noise_type='ou_noise', episodes=[4, 3])),
    ('eval_det_local', dict(D=9, A=3, ah=[16, 12], ch=[20, 16], mode='eval_deterministic_local', agent_id=1,
                            num_agents=4, noise_type='normal', episodes=[3])),
    ('eval_stoch_local_ou', dict(D=9, A=3, ah=[16, 12], ch=[20, 16], mode='eval_stochastic_local', agent_id=1,
                                 num_agents=4, noise_type='ou_noise', episodes=[2, 2])),
    ('pixel_framelist_normal', dict(D=5, A=2, ah=[16, 12], ch=[20, 16], mode='training', agent_id=1, num_agents=2,
                                    noise_type='normal', pixel=(6, 20, 24), frame_list=3, conv_hidden=8,
                                    episodes=[3])),
    ('cfg3_normal', dict(D=17, A=6, ah=[300, 200], ch=[400, 300], mode='training', agent_id=5, num_agents=8,
                         noise_type='normal', episodes=[3])),
])


def _obs(rs, D, pixel=None, frame_list=0):
    o = collections.OrderedDict()
    if pixel is not None:
        fr = rs.randint(0, 256, pixel).astype(np.uint8)
        if frame_list:      # FrameStackWrapper with frame_stack_concatenate_on_env=False (wrapper.py:454-472)
            fr = [fr[i * (pixel[0] // frame_list):(i + 1) * (pixel[0] // frame_list)] for i in range(frame_list)]
        o['pixel'] = collections.OrderedDict(camera0=fr)
    o['low_dim'] = collections.OrderedDict(flat_inputs=rs.randn(D).astype(np.float32))
    return o


def _call_recording_eps(fn, shape_of_draw):
    """run fn() and return (result, eps): the standard normals fn consumed from numpy's global
    stream (fn draws exactly one randn / normal block of `shape_of_draw`; None = no draw)"""
    st = np.random.get_state()
    out = fn()
    after = np.random.get_state()
    eps = None
    if shape_of_draw is not None:
        np.random.set_state(st)
        eps = np.random.randn(*shape_of_draw)
        mid = np.random.get_state()
        assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
                   for a, b in zip(mid, after)), 'act() drew something else than one normal block'
    return out, eps


def ppo_configs(c):
    lc = Config(PPO_DEFAULT_LEARNER_CONFIG)
    lc.model.actor_fc_hidden_sizes = list(c['hidden'])
    lc.model.critic_fc_hidden_sizes = list(c['hidden'])
    lc.model.cnn_feature_dim = c.get('cnn_feature_dim', 256)
    lc.algo.use_z_filter = c.get('use_z', True)
    rnn = c.get('rnn')
    lc.algo.rnn.if_rnn_policy = rnn is not None
    if rnn:
        lc.algo.rnn.rnn_hidden, lc.algo.rnn.rnn_layer = rnn
    ec = Config(PPO_DEFAULT_ENV_CONFIG)
    ec.action_spec = Config({'dim': [c['A']], 'type': 'continuous'})
    spec = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=[c['D']]))
    if c.get('pixel'):
        spec['pixel'] = collections.OrderedDict(camera0=list(c['pixel']))
    ec.obs_spec = spec
    ec.pixel_input = bool(c.get('pixel'))
    ec.stochastic_eval = c.get('stochastic_eval', True)
    ec.sleep_time = 0.0
    return lc, ec, Config(PPO_DEFAULT_SESSION_CONFIG)


def run_ppo_case(name, c):
    D, A = c['D'], c['A']
    rnn = c.get('rnn')
    pixel = tuple(c['pixel']) if c.get('pixel') else None
    pix_kw = dict(pixel=pixel, cnn_feature_dim=c['cnn_feature_dim']) if pixel else {}
    params = synthetic.make_ppo_params(D, A, hidden=tuple(c['hidden']), rnn_hidden=rnn[0] if rnn else 0,
                                       rnn_layers=rnn[1] if rnn else 1, **PPO_PARAM_KW, **pix_kw)
    zstate = synthetic.make_zfilter_state(D, seed=5) if c.get('use_z', True) else None
    lc, ec, sc = ppo_configs(c)
    out = {}
    n, T1, T2 = c['n_agents'], c['steps'], c.get('steps2', 0)
    T = T1 + T2
    noise = np.zeros(n)
    modes = []
    actions = np.zeros((T, n, A))
    pds = np.zeros((T, n, 2 * A), np.float32)
    eps = np.zeros((T, n, A))
    cells = np.zeros((T, n, 2, rnn[1], rnn[0]), np.float32) if rnn else None
    obs_low = np.zeros((T, n, D), np.float32)
    obs_pix = np.zeros((T, n) + pixel, np.uint8) if pixel else None
    for i in range(n):
        np.random.seed(100 + i)                      # the ctor draws the agent's exploration noise
        ag = PPOAgent(lc, ec, sc, agent_id=i, agent_mode=c['mode'])
        noise[i] = ag.noise
        modes.append(ag.agent_mode)
        GP.inject_params(ag.model, params, zstate)
        rs = np.random.RandomState(200 + i)
        np.random.seed(300 + i)                      # act's sampling stream
        stochastic = ag.agent_mode not in ('eval_deterministic', 'eval_deterministic_local')
        for t in range(T):
            if c.get('reset_after') and t == c['reset_after']:
                ag.reset()                           # episode boundary (agent/base.py:240 via main_loop)
            o = _obs(rs, D, pixel)
            obs_low[t, i] = o['low_dim']['flat_inputs']
            if pixel:
                obs_pix[t, i] = o['pixel']['camera0']
            got, e = _call_recording_eps(lambda: ag.act(o), (1, A) if stochastic else None)
            if ag.agent_mode == 'training':
                a, info = got
                assert len(info) == 2 and len(info[1]) == 1
                pds[t, i] = info[1][0]
                if rnn:
                    assert len(info[0]) == 2 and info[0][0].shape == (rnn[1], rnn[0])
                    cells[t, i, 0], cells[t, i, 1] = info[0]
                else:
                    assert info[0] == []
            else:
                a = got
                assert isinstance(a, np.ndarray) and a.shape == (A,)
            actions[t, i] = a
            if e is not None:
                eps[t, i] = e[0]
    out.update(noise=noise, actions=actions, pds=pds, eps=eps, obs_low=obs_low)
    if rnn:
        out['cells_before'] = cells
    if pixel:
        out['obs_pix'] = obs_pix
    meta = dict(c)
    meta['resolved_modes'] = modes
    meta['action_dtype'] = str(np.asarray(a).dtype)
    return out, meta


def ddpg_configs(c):
    lc = Config(DDPG_DEFAULT_LEARNER_CONFIG)
    lc.model.actor_fc_hidden_sizes = list(c['ah'])
    lc.model.critic_fc_hidden_sizes = list(c['ch'])
    lc.model.conv_spec.hidden_output_dim = c.get('conv_hidden', 200)
    lc.algo.exploration.noise_type = c['noise_type']
    ec = Config(DDPG_DEFAULT_ENV_CONFIG)
    ec.action_spec = Config({'dim': [c['A']], 'type': 'continuous'})
    spec = collections.OrderedDict()
    if c.get('pixel'):
        spec['pixel'] = collections.OrderedDict(camera0=list(c['pixel']))
    spec['low_dim'] = collections.OrderedDict(flat_inputs=[c['D']])
    ec.obs_spec = spec
    ec.pixel_input = bool(c.get('pixel'))
    ec.num_agents = c['num_agents']
    ec.sleep_time = 0.0
    ec.frame_stack_concatenate_on_env = not c.get('frame_list')
    return lc, ec, Config(DDPG_DEFAULT_SESSION_CONFIG)


def run_ddpg_case(name, c):
    import contextlib
    import io
    D, A = c['D'], c['A']
    pixel = tuple(c['pixel']) if c.get('pixel') else None
    if pixel:
        params = ddpg_oracle.make_ddpg_pixel_params(D, A, pixel, c['conv_hidden'], tuple(c['ah']), tuple(c['ch']),
                                                    seed=3)
    else:
        params = ddpg_oracle.make_ddpg_params(D, A, tuple(c['ah']), tuple(c['ch']), seed=3)
    lc, ec, sc = ddpg_configs(c)
    np.random.seed(17)
    with contextlib.redirect_stdout(io.StringIO()):     # 'Using exploration sigma ...'
        ag = DDPGAgent(lc, ec, sc, agent_id=c['agent_id'], agent_mode=c['mode'])
    GD.inject(ag.model, params)
    stochastic = ag.agent_mode not in ('eval_deterministic', 'eval_deterministic_local')
    rs = np.random.RandomState(400)
    np.random.seed(500)
    T = sum(c['episodes'])
    actions = np.zeros((T, A))
    eps = np.zeros((T, A))
    obs_low = np.zeros((T, D), np.float32)
    obs_pix = np.zeros((T,) + pixel, np.uint8) if pixel else None
    t = 0
    for ep_len in c['episodes']:
        # Agent.pre_episode (agent/base.py:198-207) only touches the parameter-fetch tracker in
        # training mode; DDPGAgent.pre_episode's own statement is the noise reset (:205-208)
        if stochastic:
            ag.noise.reset()
        for _ in range(ep_len):
            o = _obs(rs, D, pixel, c.get('frame_list', 0))
            obs_low[t] = o['low_dim']['flat_inputs']
            if pixel:
                fr = o['pixel']['camera0']
                obs_pix[t] = np.concatenate(fr, axis=0) if isinstance(fr, list) else fr
            a, e = _call_recording_eps(lambda: ag.act(o), (A,) if stochastic else None)
            actions[t] = a
            if e is not None:
                eps[t] = e
            t += 1
    out = dict(actions=actions, eps=eps, obs_low=obs_low)
    if pixel:
        out['obs_pix'] = obs_pix
    meta = dict(c)
    meta.update(sigma=float(ag.sigma), resolved_mode=ag.agent_mode, action_dtype=str(a.dtype))
    return out, meta


def noise_streams():
    out = {}
    np.random.seed(31)
    nn = AN.NormalActionNoise(np.array([0.1, -0.2, 0.0]), np.array([0.5, 1.0, 2.0]))
    out['normal'] = np.stack([nn() for _ in range(6)])
    out['normal_repr'] = repr(nn)
    np.random.seed(32)
    ou = AN.OrnsteinUhlenbeckActionNoise(mu=np.array([0.2, -0.1]), sigma=0.3, theta=0.15, dt=1e-2)
    a = [ou() for _ in range(5)]
    ou.reset()
    a += [ou() for _ in range(3)]
    out['ou'] = np.stack(a)
    np.random.seed(33)
    ou0 = AN.OrnsteinUhlenbeckActionNoise(mu=np.zeros(2), sigma=np.array([0.3, 0.6]), theta=0.5, dt=0.25,
                                          x0=np.array([1.0, -1.0]))
    out['ou_x0'] = np.stack([ou0() for _ in range(4)])
    out['ou_repr'] = repr(ou0)
    return out


def main():
    arrays, meta = {}, {'ppo': {}, 'ddpg': {}}
    for name, c in PPO_CASES.items():
        out, m = run_ppo_case(name, c)
        for k, v in out.items():
            arrays['ppo.%s.%s' % (name, k)] = v
        meta['ppo'][name] = m
        print('ppo', name, m['resolved_modes'][0], 'noise', out['noise'][:2], 'a[0,0]', out['actions'][0, 0])
    for name, c in DDPG_CASES.items():
        out, m = run_ddpg_case(name, c)
        for k, v in out.items():
            arrays['ddpg.%s.%s' % (name, k)] = v
        meta['ddpg'][name] = m
        print('ddpg', name, m['resolved_mode'], 'sigma', m['sigma'], 'a[0]', out['actions'][0])
    ns = noise_streams()
    meta['ppo_param_kw'] = PPO_PARAM_KW
    meta['noise_repr'] = {'normal': ns.pop('normal_repr'), 'ou': ns.pop('ou_repr')}
    for k, v in ns.items():
        arrays['noise.' + k] = v
    arrays['meta_json'] = np.array(json.dumps(meta))
    path = os.path.join(ROOT, 'tests', 'golden', 'agents.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
