"""CPU tier: simulator adapters, observation transforms and monitors against
tests/golden/envwrap.json, recorded from the REFERENCE's own wrappers on the scripted simulators
of tests/env_fakes.py (oracle/gen_golden_envwrap.py)."""
import collections
import json
import os

import numpy as np
import sys

import pytest

import env_fakes as F
import helpers as H
from surreal_amd import env as E
from surreal_amd.env import monitor as M
from surreal_amd.session import Config


@pytest.fixture(autouse=True)
def _no_simulator_stand_ins(monkeypatch):
    """a test that ran the reference through oracle/ref_shims.py earlier in this process (tests/test_bench_line.py) leaves
    the shims' stand-in `gym` / `robosuite` modules in sys.modules; the simulators themselves are not installed, and
    that is what these tests rely on"""
    for mod in ('gym', 'robosuite', 'dm_control'):
        m = sys.modules.get(mod)
        if m is not None and getattr(m, '__file__', None) is None:
            monkeypatch.setitem(sys.modules, mod, None)

GOLD = json.load(open(os.path.join(H.GOLDEN_DIR, 'envwrap.json')))


def plain(x):
    return json.loads(json.dumps(F.to_plain(x)))


def rt(trace):                       # run_script's output is plain already: JSON round trip only
    return json.loads(json.dumps(trace))


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


def test_gym_adapter_matches_reference():
    FakeGym = F.make_fake_gym()
    cfg = Config(pixel_input=False)
    env, cfg2 = E.wrap_gym(FakeGym(T=4), cfg)
    g = GOLD['gym']
    assert plain(env.observation_spec()) == g['obs_spec'] and plain(env.action_spec()) == g['action_spec']
    assert plain(cfg2.obs_spec) == g['obs_spec'] and plain(cfg2.action_spec) == g['action_spec']
    assert rt(F.run_script(env, 6, np.array([0.5, -0.25]))) == g['trace']
    assert plain(env.render()) == g['render']
    with pytest.raises(AssertionError):
        E.GymAdapter(FakeGym(), Config(pixel_input=True))
    disc = FakeGym()
    disc.observation_space = collections.namedtuple('Discrete', 'n shape')(3, None)
    with pytest.raises(ValueError):
        E.GymAdapter(disc, cfg).observation_spec()
    with pytest.raises(RuntimeError):                       # no double wrapping (wrapper.py:39-50)
        E.GymAdapter(env, cfg)


@pytest.mark.parametrize('name', sorted(ROBO_CASES))
def test_robosuite_stack_matches_reference(name, capsys):
    cfg = robosuite_cfg(**ROBO_CASES[name])
    env, cfg = E.wrap_robosuite(F.FakeRobosuite(T=5), cfg)
    g = GOLD['robosuite'][name]
    assert plain(env.observation_spec()) == g['obs_spec']
    assert plain(env.action_spec()) == g['action_spec'] == plain(cfg.action_spec)
    assert rt(F.run_script(env, 7, np.array([0.125, 0.25]))) == g['trace']
    assert 'skipping observation key' in capsys.readouterr().out      # the spec pass is verbose


def test_grayscale_wraps_like_the_reference():
    """np.mean(frame, 0, 'uint8') accumulates in uint8: (200 + 100 + 50) % 256 // 3, not 116"""
    class One(E.Env):
        def _reset(self):
            f = np.zeros((3, 1, 1), np.uint8)
            f[:, 0, 0] = (200, 100, 50)
            return collections.OrderedDict(pixel=collections.OrderedDict(camera0=f)), {}
    obs, _ = E.GrayscaleWrapper(One()).reset()
    assert obs['pixel']['camera0'].tolist() == [[[(350 % 256) // 3]]]


def test_make_env_categories():
    cfg = Config(env_name='synthetic:7x3', pixel_input=False)
    env, cfg = E.make_env(cfg)
    assert tuple(cfg.obs_spec['low_dim']['flat_inputs']) == (7,) and tuple(cfg.action_spec['dim']) == (3,)
    assert tuple(E.make_env_config(Config(env_name='synthetic:4x2')).action_spec['dim']) == (2,)
    with pytest.raises(ValueError):
        E.make_env(Config(env_name='atari:Pong'))
    for name in ('gym:HalfCheetah-v2', 'robosuite:SawyerLift'):   # the simulators are not installed
        with pytest.raises(ImportError):
            E.make_env(Config(env_name=name, pixel_input=False))


class Clock(object):
    def __init__(self, sim):
        self.sim = sim

    def time(self):
        return 1000.0 + 0.25 * self.sim.steps


def test_console_monitor_matches_reference(monkeypatch):
    FakeGym = F.make_fake_gym()
    sim = FakeGym(T=3)
    monkeypatch.setattr(M, 'time', Clock(sim))
    printed = []
    extra = collections.OrderedDict([('steps x episodes', lambda s, e: s * e)])
    env = E.ConsoleMonitor(E.GymAdapter(sim, Config(pixel_input=False)), update_interval=2, average_over=3,
                           extra_rows=extra, out=printed.append)
    infos = []
    env.reset()
    for i in range(13):
        _, _, done, info = env.step(np.array([1.0, float(i)]))
        if done:
            infos.append(info['episode'])
            env.reset()
    g = GOLD['console']
    assert printed == g['printed'] and infos == g['episode_infos']
    assert env.episode_rewards == g['rewards'] and env.episode_steps == g['steps']
    assert env.episode_durations == g['durations'] and env.total_steps == g['total_steps']
    assert env.num_episodes == len(g['rewards']) and env.step_per_sec(2) == g['speed2']
    with pytest.raises(AssertionError):
        E.ConsoleMonitor(E.Env(), extra_rows={'a': None})


@pytest.mark.parametrize('kind', ['training', 'eval'])
def test_tensorplex_monitors_match_reference(kind, monkeypatch):
    sess = Config(tensorplex={'update_schedule': {'training_env': 2, 'eval_env': 3, 'eval_env_sleep': 11}})
    FakeGym = F.make_fake_gym()
    sim = FakeGym(T=2)
    monkeypatch.setattr(M, 'time', Clock(sim))
    slept, fetched = [], []
    inner = E.GymAdapter(sim, Config(pixel_input=False))
    if kind == 'training':
        env = E.TrainingTensorplexMonitor(inner, 3, sess)
        assert env.tensorplex_name == 'agent/3'
        with pytest.raises(AssertionError):
            E.TrainingTensorplexMonitor(E.Env(), 'three', sess)
    else:
        env = E.EvalTensorplexMonitor(inner, 'stochastic-0', lambda: fetched.append(len(fetched)), sess,
                                      separate_plots=True, sleep=slept.append)
        assert env.tensorplex_name == 'eval/stochastic-0'
    env.reset()
    for i in range(14):
        _, _, done, _ = env.step(np.array([0.5, float(i % 3)]))
        if done:
            env.reset()
    g = GOLD[kind + '_tensorplex']
    calls = [[sorted([k, float(v)] for k, v in sc.items()), step] for step, sc in env.tensorplex.history]
    assert calls == g['calls'] and slept == g['slept'] and len(fetched) == g['fetched']


@pytest.mark.parametrize('name', ['lowdim', 'pixels_stack2'])
def test_dm_control_stack_matches_reference(name, capsys):
    FakeDM = F.make_fake_dm()
    if name == 'lowdim':
        pix, cfg = False, Config(pixel_input=False, frame_stacks=1,
                                 observation={'low_dim': ['position', 'velocity']})
    else:
        pix, cfg = True, Config(pixel_input=True, frame_stacks=2, frame_stack_concatenate_on_env=True,
                                observation={'pixel': ['camera0']})
    env, cfg = E.wrap_dm_control(FakeDM(T=4, pixels=pix), cfg)
    g = GOLD['dm_control'][name]
    assert plain(env.observation_spec()) == g['obs_spec']
    assert plain(env.action_spec()['dim']) == g['action_dim'] and env.action_spec()['type'] == 'continuous'
    assert rt(F.run_script(env, 6, np.array([0.5, 0.25]))) == g['trace']
    assert 'None reward' in capsys.readouterr().out            # the first step of an episode


def test_agent_wraps_its_env_with_the_monitors(cpu_double):
    """Agent.prepare_env (agent/base.py:283-336): step limit first, then the training monitor whose
    ':reward' / 'step_per_s' scalars arrive every update_schedule.training_env episodes; eval agents
    get the eval monitor unless they are *_local"""
    from surreal_amd.agent import PPOAgent
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    lc, ec, sc = ppo_learner_config(), ppo_env_config(5, 2), ppo_session_config('/tmp/surreal_amd_test')
    lc.algo.rnn.if_rnn_policy = False
    lc.algo.n_step, lc.algo.stride = 3, 3
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = [8, 8]
    ec.limit_episode_length = 4
    sc.tensorplex.update_schedule.training_env = 2
    sc.tensorplex.update_schedule.eval_env_sleep = 0
    ag = PPOAgent(lc, ec, sc, agent_id=1, agent_mode='training')
    ag.set_experience_sink(lambda exp: None)
    ag.set_env_factory(lambda: E.SyntheticEnv(5, 2, episode_len=50, seed=1))
    ag.main_setup()
    assert isinstance(ag.env, E.Wrapper)
    for _ in range(4):
        ag.main_loop()                                       # one episode each
    hist = ag.env_tensorplex.history
    assert [step for step, _ in hist] == [2, 4] and set(hist[0][1]) == {':reward', 'step_per_s'}
    ev = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='eval_deterministic')
    fetched = []
    ev.fetch_parameter = lambda: fetched.append(1)
    env = ev.prepare_env(E.SyntheticEnv(5, 2, seed=2))
    assert isinstance(env, E.EvalTensorplexMonitor) and fetched == [1]
    local = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='eval_deterministic_local')
    assert isinstance(local.prepare_env(E.SyntheticEnv(5, 2, seed=2)), E.MaxStepWrapper)
    with pytest.raises(ImportError):                         # no factory: make_env, and gym is absent
        ec.env_name = 'gym:HalfCheetah-v2'
        PPOAgent(lc, ec, sc, agent_id=2, agent_mode='training').get_env()
