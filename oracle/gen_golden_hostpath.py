#!/usr/bin/env python
"""
TEST INFRASTRUCTURE ONLY.  Records the behaviour of the REFERENCE's own replay buffers,
experience-windowing wrappers and aggregator (run under oracle/ref_shims.py, build container
only) on small deterministic scripts, as tests/golden/hostpath.json:

  * FIFOReplay / UniformReplay insert / sample / overflow (fifo_replay.py, uniform_replay.py)
  * ExpSenderWrapperMultiStepMovingWindowWithInfo window emission (exp_sender_wrapper.py:153-264)
  * ExpSenderWrapperSSARNStepBootstrap n-step reward accumulation incl. its ramp-up quirk (:72-112)
  * MultistepAggregatorWithInfo batch shapes (aggregator.py:106-262)
"""
import collections
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from surreal.replay.fifo_replay import FIFOReplay  # noqa: E402
from surreal.replay.uniform_replay import UniformReplay  # noqa: E402
import surreal.env.exp_sender_wrapper as esw  # noqa: E402
from surreal.learner.aggregator import MultistepAggregatorWithInfo  # noqa: E402
from surreal.session import Config  # noqa: E402


class FakeEnv(object):
    """counts steps; obs = {'low_dim': {'flat_inputs': [t, t]}}; reward = t + 1"""
    metadata = {}

    def __init__(self, T):
        self.T, self.t = T, 0

    def obs(self):
        return collections.OrderedDict(low_dim=collections.OrderedDict(
            flat_inputs=np.array([self.t, self.t], dtype=np.float32)))

    def reset(self):
        self.t = 0
        return self.obs(), {}

    def step(self, action):
        self.t += 1
        return self.obs(), float(self.t), self.t >= self.T, {}


class Capture(object):
    def __init__(self):
        self.items = []

    def send(self, hash_dict, nonhash_dict):
        d = dict(hash_dict)
        d.update(nonhash_dict)
        self.items.append(d)


def bare(cls, **attrs):
    o = object.__new__(cls)
    for k, v in attrs.items():
        setattr(o, k, v)
    return o


def main():
    out = {}
    # ---- replay ------------------------------------------------------------------------
    lc = Config({'replay': {'batch_size': 2, 'memory_size': 4, 'sampling_start_size': 2}})
    import collections as C
    f = bare(FIFOReplay, learner_config=lc, batch_size=2, memory_size=4, _memory=C.deque(maxlen=7))
    for i in range(10):
        f.insert(i)
    out['fifo_after_insert_0_9'] = list(f._memory)
    out['fifo_ready'] = bool(f.start_sample_condition())
    out['fifo_sample2'] = f.sample(2)
    out['fifo_len_after'] = len(f)
    u = bare(UniformReplay, learner_config=lc, _memory=[], memory_size=5, _next_idx=0)
    for i in range(8):
        u.insert(i)
    out['uniform_cap5_after_insert_0_7'] = list(u._memory)
    out['uniform_next_idx'] = u._next_idx
    out['uniform_ready'] = bool(u.start_sample_condition())
    random.seed(123)
    out['uniform_sample6_seed123'] = u.sample(6)
    # ---- PPO window wrapper --------------------------------------------------------------
    for T, n_step, stride in ((14, 5, 3), (10, 4, 4), (7, 3, 1), (5, 6, 2), (12, 3, 5)):
        cap = Capture()
        w = bare(esw.ExpSenderWrapperMultiStepMovingWindowWithInfo, env=FakeEnv(T), sender=cap,
                 _ob=None, n_step=n_step, stride=stride, last_n=C.deque())
        for ep in range(2):
            w._reset()
            done = False
            while not done:
                _, _, done, _ = w._step((np.zeros(1), [[], [np.array([0.5, 1.0])]]))
        out['window_T%d_n%d_s%d' % (T, n_step, stride)] = [
            {'obs_t': [int(o['low_dim']['flat_inputs'][0]) for o in e['obs']],
             'obs_next_t': int(e['obs_next']['low_dim']['flat_inputs'][0]),
             'rewards': e['rewards'], 'dones': [bool(d) for d in e['dones']], 'n_step': e['n_step']}
            for e in cap.items]
    # ---- DDPG n-step wrapper ---------------------------------------------------------------
    cap = Capture()
    w = bare(esw.ExpSenderWrapperSSARNStepBootstrap, env=FakeEnv(6), sender=cap, _obs=None,
             n_step=3, gamma=0.5, last_n=C.deque())
    w._reset()
    done = False
    while not done:
        _, _, done, _ = w._step(np.zeros(1))
    out['ssar_nstep3_gamma0.5_T6'] = [
        {'obs_t': int(e['obs'][0]['low_dim']['flat_inputs'][0]),
         'obs_next_t': int(e['obs'][1]['low_dim']['flat_inputs'][0]),
         'reward': e['reward'], 'done': bool(e['done'])} for e in cap.items]
    # ---- aggregator ------------------------------------------------------------------------
    obs_spec = {'low_dim': {'flat_inputs': [2]}}
    agg = MultistepAggregatorWithInfo(obs_spec, {'dim': [1], 'type': 'continuous'})
    cap = Capture()
    w = bare(esw.ExpSenderWrapperMultiStepMovingWindowWithInfo, env=FakeEnv(9), sender=cap,
             _ob=None, n_step=4, stride=2, last_n=C.deque())
    w._reset()
    done = False
    while not done:
        _, _, done, _ = w._step((np.array([0.25]), [[], [np.array([0.5, 1.0])]]))
    b = agg.aggregate(cap.items)
    out['aggregate_shapes'] = {
        'obs': list(b['obs']['low_dim']['flat_inputs'].shape),
        'obs_next': list(b['obs_next']['low_dim']['flat_inputs'].shape),
        'actions': list(b['actions'].shape), 'rewards': list(b['rewards'].shape),
        'dones': list(b['dones'].shape), 'dones_dtype': str(b['dones'].dtype),
        'persistent_infos': [list(x.shape) for x in b['persistent_infos']],
        'onetime_infos': b['onetime_infos'],
        'obs_first_col': b['obs']['low_dim']['flat_inputs'][:, :, 0].tolist(),
        'rewards_values': b['rewards'].tolist(),
    }
    # ---- parameter-space noise (agent/param_noise.py) ----------------------------------------
    import surreal.agent.param_noise as PN

    def params0():
        return C.OrderedDict(ddpg=C.OrderedDict([('actor.w', np.arange(6, dtype=np.float32).reshape(2, 3)),
                                                 ('actor.b', np.array([0.5, -0.5], dtype=np.float32)),
                                                 ('critic.w', np.linspace(-1, 1, 4).astype(np.float32))]))
    np.random.seed(11)
    got = PN.NormalParameterNoise(0.25).apply(params0())
    out['param_noise_normal_seed11'] = {k: np.asarray(v).tolist() for k, v in got['ddpg'].items()}

    class CleanModel(object):
        """the un-noised copy: its action is a fixed function of what was last loaded"""
        def __init__(self):
            self.loaded = None

        def __call__(self, obs, calculate_value=False):
            return np.asarray(obs, dtype=np.float64) * float(self.loaded['ddpg']['actor.b'][0]), None

    class Loader(object):
        def __init__(self, m):
            self.m = m

        def load(self, params):
            self.m.loaded = params
    clean = CleanModel()
    an = PN.AdaptiveNormalParameterNoise(clean, Loader(clean), target_stddev=0.25, compute_dist_interval=3,
                                         alpha=1.5, sigma=0.1)
    import io, contextlib
    trace = []
    with contextlib.redirect_stdout(io.StringIO()):
        np.random.seed(12)
        p = params0()
        for rnd in range(4):
            p = an.apply(p)
            trace.append({'sigma': an.sigma, 'b0': float(p['ddpg']['actor.b'][0]),
                          'clean_b0': float(clean.loaded['ddpg']['actor.b'][0])})
            for t in range(5 + rnd):
                an.compute_action_distance(np.array([1.0, 2.0]), np.array([0.1 * (t + 1) * (rnd + 1), 0.0]))
            trace[-1].update(i=an.i, dist=float(an.total_action_distance))
    out['param_noise_adaptive_seed12'] = trace
    path = os.path.join(ROOT, 'tests', 'golden', 'hostpath.json')
    with open(path, 'w') as fp:
        json.dump(out, fp, indent=1, sort_keys=True)
    print('wrote', path)
    for k, v in out.items():
        print(k, '=', v if len(str(v)) < 160 else str(v)[:160] + '...')


if __name__ == '__main__':
    main()
