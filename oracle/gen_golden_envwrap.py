#!/usr/bin/env python
"""
TEST INFRASTRUCTURE ONLY.  Records what the REFERENCE's env adapters, observation transforms and
monitors (surreal/env/wrapper.py:165-513, make_env.py:93-104, monitor.py:11-218; run under
oracle/ref_shims.py, build container only) do on the scripted simulators of tests/env_fakes.py,
as tests/golden/envwrap.json.  tests/test_env_adapters.py replays the same scripts on
surreal_amd.env.
"""
import collections
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_shims  # noqa: E402

ref_shims.install()
import gym  # noqa: E402  (the shim)
import types  # noqa: E402
import env_fakes as F  # noqa: E402

# dm_control is absent too: just enough of its module tree for surreal/env/dm_wrapper.py to import
_dm = types.ModuleType('dm_control')
_dm.rl = types.ModuleType('dm_control.rl')
_dm.rl.control = types.ModuleType('dm_control.rl.control')
_dm.rl.control.Environment = type('Environment', (object,), {})
_dm.rl.environment = types.ModuleType('dm_control.rl.environment')
_dm.rl.environment.StepType = F.StepType
_dm.rl.specs = types.ModuleType('dm_control.rl.specs')
_dm.rl.specs.ArraySpec = F.ArraySpec
_dm.suite = types.ModuleType('dm_control.suite')
_dm.suite.wrappers = types.ModuleType('dm_control.suite.wrappers')
_dm.suite.wrappers.pixels = types.ModuleType('dm_control.suite.wrappers.pixels')
_dm.suite.wrappers.pixels.Wrapper = type('Wrapper', (object,), {})
for _m in (_dm, _dm.rl, _dm.rl.control, _dm.rl.environment, _dm.rl.specs, _dm.suite, _dm.suite.wrappers,
           _dm.suite.wrappers.pixels):
    sys.modules[_m.__name__] = _m
import surreal.env.wrapper as W  # noqa: E402
import surreal.env.monitor as M  # noqa: E402
from surreal.session import Config  # noqa: E402
import surreal.env.dm_wrapper as DMW  # noqa: E402


def robosuite_stack(sim, cfg):
    """make_env.py:93-104 around an already built simulator"""
    env = W.RobosuiteWrapper(sim, cfg)
    env = W.FilterWrapper(env, cfg)
    env = W.ObservationConcatenationWrapper(env)
    if cfg.pixel_input:
        env = W.TransposeWrapper(env)
        if cfg.use_grayscale:
            env = W.GrayscaleWrapper(env)
        if cfg.frame_stacks:
            env = W.FrameStackWrapper(env, cfg)
    return env


def robosuite_cfg(**kw):
    base = dict(pixel_input=True, use_depth=False, use_grayscale=False, frame_stacks=0,
                frame_stack_concatenate_on_env=True, action_repeat=1,
                observation={'pixel': ['camera0'], 'low_dim': ['robot-state', 'object-state']})
    base.update(kw)
    return Config(base)


ROBO_CASES = {
    'depth_repeat2': dict(use_depth=True, action_repeat=2),
    'gray_stack3': dict(use_grayscale=True, frame_stacks=3),
    'lowdim_only': dict(pixel_input=False, observation={'pixel': [], 'low_dim': ['object-state']}),
}


class Clock(object):
    """wall clock driven by the simulator's step counter (0.25 s per env step)"""

    def __init__(self, sim):
        self.sim = sim

    def time(self):
        return 1000.0 + 0.25 * self.sim.steps


class Capture(object):
    def __init__(self):
        self.calls = []

    def add_scalars(self, scalars, global_step=None):
        self.calls.append([sorted([k, float(v)] for k, v in scalars.items()), global_step])


def main():
    out = {}
    # ---- Gym ---------------------------------------------------------------------------------
    FakeGym = F.make_fake_gym(base=gym.Env, box=gym.spaces.Box)
    env = W.GymAdapter(FakeGym(T=4), Config(pixel_input=False))
    out['gym'] = {'obs_spec': F.to_plain(env.observation_spec()), 'action_spec': F.to_plain(env.action_spec()),
                  'trace': F.run_script(env, 6, np.array([0.5, -0.25])),
                  'render': F.to_plain(env.render())}
    # ---- Robosuite stacks --------------------------------------------------------------------
    out['robosuite'] = {}
    for name, kw in ROBO_CASES.items():
        cfg = robosuite_cfg(**kw)
        env = robosuite_stack(F.FakeRobosuite(T=5), cfg)
        out['robosuite'][name] = {'obs_spec': F.to_plain(env.observation_spec()),
                                  'action_spec': F.to_plain(env.action_spec()),
                                  'trace': F.run_script(env, 7, np.array([0.125, 0.25]))}
    # ---- dm_control stacks (make_env.py:125-135) ---------------------------------------------
    FakeDM = F.make_fake_dm(base=_dm.rl.control.Environment)
    out['dm_control'] = {}
    for name, pix, cfg in (
            ('lowdim', False, Config(pixel_input=False, frame_stacks=1,
                                     observation={'low_dim': ['position', 'velocity']})),
            ('pixels_stack2', True, Config(pixel_input=True, frame_stacks=2, frame_stack_concatenate_on_env=True,
                                           observation={'pixel': ['camera0']}))):
        env = DMW.DMControlAdapter(FakeDM(T=4, pixels=pix), pix)
        env = W.FilterWrapper(env, cfg)
        env = W.ObservationConcatenationWrapper(env)
        if pix:
            env = W.GrayscaleWrapper(W.TransposeWrapper(env))
            if cfg.frame_stacks > 1:
                env = W.FrameStackWrapper(env, cfg)
        out['dm_control'][name] = {'obs_spec': F.to_plain(env.observation_spec()),
                                   'action_dim': F.to_plain(env.action_spec()['dim']),
                                   'trace': F.run_script(env, 6, np.array([0.5, 0.25]))}
    # ---- monitors ----------------------------------------------------------------------------
    sim = FakeGym(T=3)
    M.time = Clock(sim)                     # monitor.py reads time.time() only
    printed = []
    M.print = lambda *a: printed.append(' '.join(str(x) for x in a))
    extra = collections.OrderedDict([('steps x episodes', lambda s, e: s * e)])
    env = M.ConsoleMonitor(W.GymAdapter(sim, Config(pixel_input=False)), update_interval=2, average_over=3,
                           extra_rows=extra)
    infos = []
    env.reset()
    for i in range(13):
        _, _, done, info = env.step(np.array([1.0, float(i)]))
        if done:
            infos.append(info['episode'])
            env.reset()
    out['console'] = {'printed': printed, 'episode_infos': infos, 'rewards': env.episode_rewards,
                      'steps': env.episode_steps, 'durations': env.episode_durations,
                      'total_steps': env.total_steps, 'speed2': env.step_per_sec(2)}
    M.get_tensorplex_client = lambda name, session_config: Capture()   # no tensorplex server here
    sess = Config(tensorplex={'update_schedule': {'training_env': 2, 'eval_env': 3, 'eval_env_sleep': 11}})
    for kind in ('training', 'eval'):
        sim = FakeGym(T=2)
        M.time = Clock(sim)
        slept, fetched = [], []
        M.time.sleep = slept.append
        inner = W.GymAdapter(sim, Config(pixel_input=False))
        if kind == 'training':
            env = M.TrainingTensorplexMonitor(inner, 3, sess)
        else:
            env = M.EvalTensorplexMonitor(inner, 'stochastic-0', lambda: fetched.append(len(fetched)), sess,
                                          separate_plots=True)
        cap = env.tensorplex = Capture()
        env.reset()
        for i in range(14):
            _, _, done, _ = env.step(np.array([0.5, float(i % 3)]))
            if done:
                env.reset()
        out[kind + '_tensorplex'] = {'calls': cap.calls, 'slept': slept, 'fetched': len(fetched)}
    path = os.path.join(ROOT, 'tests', 'golden', 'envwrap.json')
    json.dump(out, open(path, 'w'), indent=0, sort_keys=True)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
