"""CPU tier: replay buffers, experience windowing, aggregation and the agent loop of the product
against tests/golden/hostpath.json, recorded from the REFERENCE's own classes
(oracle/gen_golden_hostpath.py), plus the device tier of the same objects through the CPU
kernel double, plus the in-process Agent -> Replay -> Learner loop (the reference's only live
test, test/test_ppo_gym.py -> test_helpers/integration_test.py:42-105, asserts "no exception";
this one also checks what flowed)."""
import collections
import json
import os
import random
import sys

import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import pytest
import torch

import helpers as H
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config

GOLD = json.load(open(os.path.join(H.GOLDEN_DIR, 'hostpath.json')))


class FakeEnv(object):
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


def configs(B=2, N=4, stride=2, D=2, A=1, memory=4):
    lc = ppo_learner_config()
    lc.algo.n_step, lc.algo.stride = N, stride
    lc.algo.rnn.if_rnn_policy = False
    lc.replay.batch_size, lc.replay.memory_size, lc.replay.sampling_start_size = B, memory, B
    lc.model.actor_fc_hidden_sizes = [24, 16]
    lc.model.critic_fc_hidden_sizes = [24, 16]
    return lc, ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test')


def test_fifo_replay_matches_reference():
    from surreal_amd.replay import FIFOReplay
    lc, ec, sc = configs()
    f = FIFOReplay(lc, ec, sc)
    for i in range(10):
        f.insert(i)
    assert list(f._memory) == GOLD['fifo_after_insert_0_9']          # maxlen = memory_size + 3
    assert f.start_sample_condition() == GOLD['fifo_ready']
    assert f.sample(2) == GOLD['fifo_sample2'] and len(f) == GOLD['fifo_len_after']
    with pytest.raises(NotImplementedError):
        f.evict()
    empty = FIFOReplay(lc, ec, sc)
    assert not empty.start_sample_condition() and len(empty) == 0
    assert empty._sample_request_handler(2) is None


def test_uniform_replay_matches_reference():
    from surreal_amd.replay import UniformReplay
    lc, ec, sc = configs(memory=5)
    u = UniformReplay(lc, ec, sc)
    for i in range(8):
        u.insert(i)
    assert list(u._memory) == GOLD['uniform_cap5_after_insert_0_7']
    assert u._next_idx == GOLD['uniform_next_idx'] and u.start_sample_condition()
    random.seed(123)
    assert u.sample(6) == GOLD['uniform_sample6_seed123']             # with replacement, same stream


@pytest.mark.parametrize('T,n_step,stride', [(14, 5, 3), (10, 4, 4), (7, 3, 1), (5, 6, 2), (12, 3, 5)])
def test_ppo_window_wrapper_matches_reference(T, n_step, stride):
    from surreal_amd.env import ExpSenderWrapperMultiStepMovingWindowWithInfo
    lc, ec, sc = configs(N=n_step, stride=stride)
    got = []
    w = ExpSenderWrapperMultiStepMovingWindowWithInfo(FakeEnv(T), lc, sc, sink=got.append)
    for ep in range(2):
        w.reset()
        done = False
        while not done:
            _, _, done, _ = w.step((np.zeros(1), [[], [np.array([0.5, 1.0])]]))
    ref = GOLD['window_T%d_n%d_s%d' % (T, n_step, stride)]
    from surreal_amd.env.exp_sender_wrapper import windows_per_episode
    # (12, 3, 5): a stride past the window length advances by n_step -- the reference pops `stride`
    # entries off a queue that holds only n_step
    assert len(got) == len(ref) == 2 * windows_per_episode(T, n_step, stride)
    for e, r in zip(got, ref):
        assert [int(o['low_dim']['flat_inputs'][0]) for o in e['obs']] == r['obs_t']
        assert int(e['obs_next']['low_dim']['flat_inputs'][0]) == r['obs_next_t']
        assert e['rewards'] == r['rewards'] and [bool(d) for d in e['dones']] == r['dones']
        assert e['n_step'] == r['n_step'] and e['onetime_infos'] == []


def test_ddpg_nstep_wrapper_matches_reference_quirks():
    from surreal_amd.env import ExpSenderWrapperSSARNStepBootstrap
    lc, ec, sc = configs(N=3)
    lc.algo.gamma = 0.5
    got = []
    w = ExpSenderWrapperSSARNStepBootstrap(FakeEnv(6), lc, sc, sink=got.append)
    w.reset()
    done = False
    while not done:
        _, _, done, _ = w.step(np.zeros(1))
    ref = GOLD['ssar_nstep3_gamma0.5_T6']
    assert len(got) == len(ref)
    for e, r in zip(got, ref):
        assert int(e['obs'][0]['low_dim']['flat_inputs'][0]) == r['obs_t']
        assert int(e['obs'][1]['low_dim']['flat_inputs'][0]) == r['obs_next_t']
        assert e['reward'] == r['reward'] and bool(e['done']) == r['done']


def test_aggregator_matches_reference():
    from surreal_amd.env import ExpSenderWrapperMultiStepMovingWindowWithInfo
    from surreal_amd.learner.aggregator import MultistepAggregatorWithInfo
    lc, ec, sc = configs(N=4, stride=2)
    got = []
    w = ExpSenderWrapperMultiStepMovingWindowWithInfo(FakeEnv(9), lc, sc, sink=got.append)
    w.reset()
    done = False
    while not done:
        _, _, done, _ = w.step((np.array([0.25]), [[], [np.array([0.5, 1.0])]]))
    b = MultistepAggregatorWithInfo(ec.obs_spec, ec.action_spec).aggregate(got)
    r = GOLD['aggregate_shapes']
    assert list(b['obs']['low_dim']['flat_inputs'].shape) == r['obs']
    assert list(b['obs_next']['low_dim']['flat_inputs'].shape) == r['obs_next']
    assert list(b['actions'].shape) == r['actions'] and list(b['rewards'].shape) == r['rewards']
    assert list(b['dones'].shape) == r['dones'] and str(b['dones'].dtype) == r['dones_dtype']
    assert [list(x.shape) for x in b['persistent_infos']] == r['persistent_infos']
    assert b['onetime_infos'] is r['onetime_infos'] is None
    assert b['obs']['low_dim']['flat_inputs'][:, :, 0].tolist() == r['obs_first_col']
    assert b['rewards'].tolist() == r['rewards_values']
    with pytest.raises(ValueError):
        MultistepAggregatorWithInfo(ec.obs_spec, ec.action_spec).aggregate(
            [got[0], dict(got[1], actions=got[1]['actions'][:2])])      # ragged -> loud


def _host_batch_out(B, N, D, A, pixel=None, hid=None):
    out = {'obs': {'low_dim': {'flat_inputs': np.full((B, N, D), np.nan, np.float32)}},
           'obs_next': {'low_dim': {'flat_inputs': np.full((B, 1, D), np.nan, np.float32)}},
           'actions': np.full((B, N, A), np.nan, np.float32), 'rewards': np.full((B, N), np.nan, np.float32),
           'dones': np.full((B, N), np.nan, np.float32), 'persistent_infos': [np.full((B, N, 2 * A), np.nan, np.float32)],
           'onetime_infos': None if hid is None else [np.full((B, 1, hid), np.nan, np.float32) for _ in range(2)]}
    if pixel:
        out['obs']['pixel'] = {'camera0': np.zeros((B, N) + pixel, np.uint8)}
        out['obs_next']['pixel'] = {'camera0': np.zeros((B, 1) + pixel, np.uint8)}
    return out


@pytest.mark.parametrize('layout', ['float32', 'float64_obs', 'pixel_rnn', 'array_fields', 'unsupported_leaf'])
def test_native_batch_assembly_equals_the_numpy_path(layout, monkeypatch):
    """MultistepAggregatorWithInfo.aggregate(out=pinned staging views): the CPython extension (csrc/host/smx_host.c --
    leaf pointers collected under the GIL, copies / float64 -> float32 / bool -> float32 conversions done without it)
    writes the same bytes as the numpy path for what the agents send (lists of per-step arrays, Python floats and
    bools, uint8 camera frames, LSTM states as one-time infos, whole (N, ...) arrays per field), and hands a field
    with a leaf it does not know (float16) to numpy instead of guessing"""
    from surreal_amd.learner import aggregator as AG
    assert AG.native_fill() is not None, 'build it: python -m surreal_amd.build'
    B, N, D, A = 6, 5, 7, 3
    pixel = (2, 6, 4) if layout == 'pixel_rnn' else None
    hid = 4 if layout == 'pixel_rnn' else None
    rs = np.random.RandomState(5)
    odt = np.float64 if layout == 'float64_obs' else (np.float16 if layout == 'unsupported_leaf' else np.float32)

    def ob():
        o = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=rs.randn(D).astype(odt)))
        if pixel:
            o['pixel'] = collections.OrderedDict(camera0=rs.randint(0, 256, pixel).astype(np.uint8))
        return o
    exps = []
    for _ in range(B):
        e = {'obs': [ob() for _ in range(N)], 'obs_next': ob(),
             'actions': [rs.randn(A).astype(np.float32) for _ in range(N)],
             'rewards': [float(rs.randn()) for _ in range(N)], 'dones': [bool(rs.rand() < 0.3) for _ in range(N)],
             'persistent_infos': [[rs.randn(2 * A).astype(np.float32)] for _ in range(N)],
             'onetime_infos': [] if hid is None else [rs.randn(1, hid).astype(np.float32), rs.randn(1, hid).astype(np.float32)]}
        if layout == 'array_fields':                 # one array per field (what a device-tier replay hands over)
            e['actions'] = np.stack(e['actions'])
            e['rewards'] = np.asarray(e['rewards'], np.float64)
            e['dones'] = np.asarray(e['dones'])
        exps.append(e)
    spec = {'low_dim': {'flat_inputs': [D]}}
    if pixel:
        spec['pixel'] = {'camera0': list(pixel)}
    agg = AG.MultistepAggregatorWithInfo(spec, {'type': 'continuous', 'dim': [A]})
    native = agg.aggregate(exps, out=_host_batch_out(B, N, D, A, pixel, hid))
    calls = []
    real = AG.native_fill()
    monkeypatch.setattr(AG, '_NATIVE', [True, lambda *a: calls.append(a[2]) or real(*a)])
    agg.aggregate(exps, out=_host_batch_out(B, N, D, A, pixel, hid))
    assert 'obs' in calls and 'rewards' in calls and 'persistent_infos' in calls      # the extension is what ran
    monkeypatch.setattr(AG, '_NATIVE', [True, None])
    plain = agg.aggregate(exps, out=_host_batch_out(B, N, D, A, pixel, hid))

    def same(a, b, where=''):
        if isinstance(a, dict):
            assert list(a) == list(b)
            for k in a:
                same(a[k], b[k], where + '/' + str(k))
        elif isinstance(a, list):
            assert len(a) == len(b)
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, where + '[%d]' % i)
        elif a is None:
            assert b is None, where
        else:
            assert a.dtype == b.dtype and not np.isnan(a.astype(np.float64)).any(), where
            assert np.array_equal(a, b), where
    same(native, plain)


def test_native_batch_assembly_refuses_what_it_cannot_place():
    from surreal_amd.learner import aggregator as AG
    fill = AG.native_fill()
    exps = [{'x': [np.zeros(3, np.float32), np.zeros(3, np.float32)]} for _ in range(2)]
    assert fill(np.zeros((2, 2, 3), np.float32), exps, 'x', ()) == 4
    assert fill(np.zeros((2, 2, 4), np.float32), exps, 'x', ()) == -1           # slot size
    assert fill(np.zeros((3, 2, 3), np.float32), exps, 'x', ()) == -1           # batch size
    assert fill(np.zeros((2, 3, 3), np.float32), exps, 'x', ()) == -1           # steps
    assert fill(np.zeros((2, 2, 3), np.float64), exps, 'x', ()) == -1           # staging is float32 / uint8
    assert fill(np.zeros((2, 2, 3), np.float32)[:, :, ::1].transpose(1, 0, 2), exps, 'x', ()) == -1   # not contiguous
    assert fill(np.zeros((2, 2, 3), np.float32), exps, 'y', ()) == -1           # no such field
    out = np.full((2, 2, 3), 7.0, np.float32)
    assert fill(out, [exps[0], {'x': [np.zeros(3, np.float32), np.zeros(3, np.float16)]}], 'x', ()) == -1
    assert float(out.min()) == 7.0                                               # nothing written on refusal


def test_maxstep_and_framestack_wrappers():
    from surreal_amd.env import MaxStepWrapper, FrameStackWrapper
    from surreal_amd.session import Config
    env = MaxStepWrapper(FakeEnv(100), 3)
    env.reset()
    dones = [env.step(0)[2] for _ in range(3)]
    assert dones == [False, False, True]
    with pytest.raises(RuntimeError):
        MaxStepWrapper(env, 3)                                        # no double wrapping

    class Pix(FakeEnv):
        def obs(self):
            return collections.OrderedDict(pixel=collections.OrderedDict(
                camera0=np.full((1, 2, 2), self.t, dtype=np.uint8)))
    fs = FrameStackWrapper(Pix(10), Config(frame_stacks=3, frame_stack_concatenate_on_env=True))
    o, _ = fs.reset()
    assert o['pixel']['camera0'].shape == (3, 2, 2) and o['pixel']['camera0'][:, 0, 0].tolist() == [0, 0, 0]
    o = fs.step(0)[0]
    o = fs.step(0)[0]
    assert o['pixel']['camera0'][:, 0, 0].tolist() == [0, 1, 2]


def test_device_tier_replay_and_vec_env(cpu_double):
    """SoA tables + ring insert / FIFO pop / uniform gather; vectorised env + window emission
    against the single-actor numpy env driven through the reference-style wrapper"""
    from surreal_amd.replay import FIFOReplay, UniformReplay
    from surreal_amd.env import SyntheticEnv, SyntheticVecEnv, ExpSenderWrapperMultiStepMovingWindowWithInfo
    n, D, A, T, n_step, stride = 3, 7, 2, 11, 4, 3
    g = torch.Generator().manual_seed(0)
    acts = torch.randn(T, n, A, generator=g) * 0.7
    venv = SyntheticVecEnv(n, D, A, episode_len=T, seeds=[5, 6, 7])
    venv.start_rollout(T, info_width=2 * A)
    for t in range(T):
        venv.step(acts[t], pds=torch.full((n, 2 * A), float(t)))
    win = venv.emit_windows(n_step, stride)
    W = (T - n_step) // stride + 1
    assert win['obs'].shape == (n * W, n_step, D) and win['obs_next'].shape == (n * W, 1, D)
    lc, ec, sc = configs(N=n_step, stride=stride, D=D, A=A)
    for a in range(n):                                     # same thing, one actor at a time
        got = []
        w = ExpSenderWrapperMultiStepMovingWindowWithInfo(SyntheticEnv(D, A, T, seed=5 + a), lc, sc,
                                                          sink=got.append)
        w.reset()
        for t in range(T):
            w.step((acts[t, a].numpy(), [[], [np.zeros(2 * A)]]))
        assert len(got) == W
        for k, e in enumerate(got):
            ob = np.stack([o['low_dim']['flat_inputs'] for o in e['obs']])
            np.testing.assert_array_equal(win['obs'][a * W + k].numpy(), ob)
            np.testing.assert_array_equal(win['obs_next'][a * W + k, 0].numpy(),
                                          e['obs_next']['low_dim']['flat_inputs'])
            np.testing.assert_array_equal(win['rewards'][a * W + k].numpy(),
                                          np.array(e['rewards'], dtype=np.float32))
            np.testing.assert_array_equal(win['dones'][a * W + k].numpy(),
                                          np.array(e['dones'], dtype=np.float32))
            np.testing.assert_array_equal(win['actions'][a * W + k].numpy(),
                                          np.clip(np.stack(e['actions']), -1, 1).astype(np.float32))
    # FIFO device tier: conveyor semantics incl. overflow (capacity memory_size + 3)
    lc.replay.memory_size, lc.replay.batch_size = 6, 4
    f = FIFOReplay(lc, ec, sc)
    fields = {k: v for k, v in win.items()}
    f.insert_batch(fields)                                 # 9 experiences into capacity 9
    assert len(f) == n * W and f.start_sample_condition()
    b = f.sample_batch(4)
    assert torch.equal(b['obs'], win['obs'][:4]) and len(f) == n * W - 4
    f.insert_batch({k: v[:6] for k, v in fields.items()})  # 5 + 6 = 11 > 9: two oldest dropped
    assert len(f) == 9
    b = f.sample_batch(4)
    assert torch.equal(b['rewards'], torch.cat([win['rewards'][6:9], win['rewards'][:1]]))
    # uniform device tier: ring overwrite + injected indices
    lc.replay.memory_size = 5
    u = UniformReplay(lc, ec, sc)
    u.insert_batch({k: v[:8] for k, v in fields.items()})  # capacity 5 <- last 5 of 8... ring
    assert len(u) == 5
    u2 = UniformReplay(lc, ec, sc)
    for i in range(8):
        u2.insert_batch({k: v[i:i + 1] for k, v in fields.items()})
    got = u2.sample_batch(5, indices=[0, 1, 2, 3, 4])
    np.testing.assert_array_equal(got['obs'].numpy(), win['obs'][[5, 6, 7, 3, 4]].numpy())
    idx = u2.sample_indices(64)
    assert int(idx.min()) >= 0 and int(idx.max()) < 5


def test_device_tier_keeps_uint8_frames(cpu_double):
    """host logic of the per-field table dtype: uint8 fields get uint8 tables, everything
    else is widened to fp32; reserve_batch hands out views of the right dtype"""
    from surreal_amd.replay import FIFOReplay, UniformReplay
    from surreal_amd.replay.base import table_dtype
    assert table_dtype(torch.uint8) == torch.uint8 and table_dtype(torch.float64) == torch.float32
    lc, ec, sc = configs(N=4, D=6, A=2)
    lc.replay.memory_size, lc.replay.batch_size = 6, 2
    g = torch.Generator().manual_seed(1)
    frames = torch.randint(0, 255, (5, 4, 3, 6, 6), generator=g).to(torch.uint8)
    dones = torch.randint(0, 2, (5, 4), generator=g)                 # int64 -> fp32 like the batch contract
    f = FIFOReplay(lc, ec, sc)
    f.insert_batch({'frames': frames, 'dones': dones})
    assert f._tables['frames'].data.dtype == torch.uint8 and f._tables['dones'].data.dtype == torch.float32
    b = f.sample_batch(3)
    assert b['frames'].dtype == torch.uint8 and torch.equal(b['frames'], frames[:3])
    assert torch.equal(b['dones'], dones[:3].float())
    views = f.reserve_batch(2, {'frames': (4, 3, 6, 6), 'dones': (4,)}, dtypes={'frames': torch.uint8})
    assert views['frames'].dtype == torch.uint8 and views['frames'].shape == (2, 4, 3, 6, 6)
    u = UniformReplay(lc, ec, sc)
    u.insert_batch({'frames': frames})
    assert torch.equal(u.sample_batch(2, indices=[4, 1])['frames'], frames[[4, 1]])


def test_agent_replay_learner_loop_in_process(cpu_double):
    """agents (reference-style, one env each) -> windowing wrapper -> FIFO replay -> learner;
    then the learner publishes and the agents pick the new parameters up."""
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticEnv
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    D, A, N = 7, 2, 5
    lc, ec, sc = configs(B=4, N=N, stride=3, D=D, A=A, memory=8)
    ec.limit_episode_length = 12
    lc.parameter_publish.exp_interval = 4
    lc.parameter_publish.min_publish_interval = 0.0
    replay = FIFOReplay(lc, ec, sc)
    learner = PPOLearner(lc, ec, sc)
    learner.attach_replay(replay)
    agents = []
    for i in range(2):
        ag = PPOAgent(lc, ec, sc, agent_id=i, agent_mode='training')
        ag.set_experience_sink(replay._insert_wrapper)
        ag.set_env_factory(lambda i=i: SyntheticEnv(D, A, episode_len=50, seed=i))
        ag.attach_learner(learner)
        ag.main_setup()
        agents.append(ag)
    learner.main_setup()
    for it in range(2):
        for ag in agents:
            ag.main_loop()
            ag.main_loop()
        assert replay.start_sample_condition()
        learner.main_loop()
    assert learner.current_iter == 2 and learner.epochs_executed >= 1
    assert replay.cumulative_collected_count == 2 * 2 * 2 * ((12 - N) // 3 + 1)
    st = learner.tensorplex.latest
    for k in ('_surr_loss', '_val_loss', '_pol_kl', '_entropy', '_avg_return_targ', 'obs_running_mean'):
        assert np.isfinite(st[k]), k
    # publish happened (exp_interval reached) -> reference policy refreshed, agents can fetch
    assert learner.exp_counter == 0
    before = agents[0].model.actor_flat.clone()
    assert agents[0].fetch_parameter() and not agents[0].fetch_parameter()
    assert torch.equal(agents[0].model.actor_flat, learner.model.actor_flat)
    assert not torch.equal(before, agents[0].model.actor_flat)
    # eval agent: deterministic action = clipped mean
    ev = PPOAgent(lc, ec, sc, agent_id=9, agent_mode='eval_deterministic_local')
    o, _ = SyntheticEnv(D, A, seed=3).reset()
    a1, a2 = ev.act(o), ev.act(o)
    np.testing.assert_array_equal(a1, a2)
    acts, pds = agents[0].act_batch(torch.randn(6, D), eps=torch.zeros(6, A))
    np.testing.assert_allclose(acts.numpy(), np.clip(pds[:, :A].numpy(), -1, 1))


@pytest.mark.parametrize('LAYERS', [1, 2])
def test_rnn_agent_replay_learner_loop_in_process(cpu_double, LAYERS):
    """the reference's DEFAULT policy (LSTM stem; also with stacked layers, rnn_layer = 2): the agent carries (h, c) across steps, every
    window's onetime_infos hold the state before its first step (ppo_agent.py:133-135,
    exp_sender_wrapper.py:237-252), and the learner's sequence pass from that state reproduces
    the per-step policies the agent acted with -- the property the whole RNN learner rests on."""
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticEnv
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    D, A, N, HID = 5, 2, 6, 12
    lc, ec, sc = configs(B=3, N=N, stride=2, D=D, A=A, memory=16)
    lc.algo.rnn.if_rnn_policy = True
    lc.algo.rnn.rnn_hidden = HID
    lc.algo.rnn.rnn_layer = LAYERS
    lc.algo.rnn.horizon = 3
    ec.limit_episode_length = 14
    replay = FIFOReplay(lc, ec, sc)
    learner = PPOLearner(lc, ec, sc)
    learner.attach_replay(replay)
    ag = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='eval_stochastic_local')
    assert ag.cells[0].shape == (LAYERS, 1, HID) and float(ag.cells[0].abs().sum()) == 0.0
    ag = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='training')
    ag.noise = 0.0                                   # compare pds without the exploration scale
    collected = []
    ag.set_experience_sink(lambda exp: (collected.append(exp), replay._insert_wrapper(exp)))
    ag.set_env_factory(lambda: SyntheticEnv(D, A, episode_len=50, seed=4))
    ag.attach_learner(learner)
    ag.main_setup()
    learner.main_setup()
    ag.main_loop()
    assert len(collected) == (14 - N) // 2 + 1
    for exp in collected:
        h, c = exp['onetime_infos']
        assert h.shape == (LAYERS, HID) and c.shape == (LAYERS, HID)
    assert float(np.abs(collected[0]['onetime_infos'][0]).sum()) == 0.0      # episode start
    assert float(np.abs(collected[1]['onetime_infos'][0]).sum()) > 0.0
    # sequence pass from the stored state == the step-by-step policies the agent produced
    exp = collected[1]
    obs_seq = torch.as_tensor(np.stack([o['low_dim']['flat_inputs'] for o in exp['obs'][:N]]))[None]
    cells = tuple(torch.as_tensor(x).view(LAYERS, 1, HID) for x in exp['onetime_infos'])
    pd_seq = ag.model.forward_actor({'low_dim': {'flat_inputs': obs_seq}}, cells)[0].numpy()
    pd_steps = np.stack(exp['persistent_infos'][0]) if isinstance(exp['persistent_infos'], list) \
        and len(exp['persistent_infos']) == 1 else np.stack([p[-1] for p in exp['persistent_infos']])
    np.testing.assert_allclose(pd_seq, pd_steps.reshape(pd_seq.shape), atol=1e-6)
    # learner consumes the windows (onetime_infos -> (1, B, H) cells)
    ag.main_loop()
    assert replay.start_sample_condition()
    learner.main_loop()
    st = learner.tensorplex.latest
    for k in ('_surr_loss', '_val_loss', '_pol_kl', 'grad_norm_actor', 'grad_norm_critic'):
        assert np.isfinite(st[k]), k
    assert learner._ws.E == N - 3 + 1 and learner._ws.rows == 3 * (N - 3 + 1)
    # batched acting: per-actor state, reset by mask
    acts, pds = ag.act_batch(torch.randn(4, D), eps=torch.zeros(4, A))
    assert float(ag.batch_cells_before[0].abs().sum()) == 0.0
    acts, pds = ag.act_batch(torch.randn(4, D), eps=torch.zeros(4, A))
    assert float(ag.batch_cells_before[0].abs().sum()) > 0.0
    ag.reset_batch(torch.tensor([True, False, False, True]))
    hb = ag._batch_cells[0][0]
    assert float(hb[0].abs().sum()) == 0.0 and float(hb[1].abs().sum()) > 0.0


def test_pixel_agent_replay_learner_loop_in_process(cpu_double):
    """camera-frame observations (cfg 4's shape of problem): uint8 frames travel agent -> window
    wrapper -> replay -> aggregator unchanged, the learner keeps them uint8 on the device, and
    the CNN stem + LSTM stem policy trains on them."""
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticEnv
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    D, A, N, CAM = 4, 2, 5, (2, 20, 24)
    lc, _, sc = configs(B=3, N=N, stride=2, D=D, A=A, memory=16)
    ec = ppo_env_config(D, A, pixel=CAM)
    ec.limit_episode_length = 11
    lc.model.cnn_feature_dim = 8
    lc.algo.rnn.if_rnn_policy = True
    lc.algo.rnn.rnn_hidden = 12
    lc.algo.rnn.horizon = 2
    replay = FIFOReplay(lc, ec, sc)
    learner = PPOLearner(lc, ec, sc)
    assert learner.model.if_pixel and learner.model.stem_in == D + 8
    assert learner.model.actor_flat.numel() + learner.model.critic.numel == learner.model.flat.numel()
    learner.attach_replay(replay)
    ag = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='training')
    collected = []
    ag.set_experience_sink(lambda exp: (collected.append(exp), replay._insert_wrapper(exp)))
    ag.set_env_factory(lambda: SyntheticEnv(D, A, episode_len=50, seed=2, pixel=CAM))
    ag.attach_learner(learner)
    ag.main_setup()
    learner.main_setup()
    ag.main_loop()
    assert len(collected) == (11 - N) // 2 + 1
    fr = collected[0]['obs'][0]['pixel']['camera0']
    assert fr.dtype == np.uint8 and fr.shape == CAM
    batch = learner.aggregator.aggregate(collected[:3])
    assert batch['obs']['pixel']['camera0'].shape == (3, N) + CAM
    assert batch['obs']['pixel']['camera0'].dtype == np.uint8
    assert batch['obs_next']['pixel']['camera0'].shape == (3, 1) + CAM
    learner.main_loop()
    st = learner.tensorplex.latest
    for k in ('_surr_loss', '_val_loss', '_pol_kl', 'grad_norm_actor', 'grad_norm_critic'):
        assert np.isfinite(st[k]), k
    assert learner._ws.frames_it.dtype == torch.uint8          # never widened to fp32
    # the stems were trained by both optimisers
    m = learner.model
    assert float(learner.actor_exp_avg[m.n_actor_block:].abs().sum()) > 0
    assert float(learner.critic_exp_avg[:m.n_stem].abs().sum()) > 0


def test_fifo_zero_copy_insert_and_pop_equal_the_copy_path(cpu_double):
    """reserve_batch / commit_batch (producer writes the table rows in place) and
    sample_batch(copy=False) (contiguous pops come back as views) hold the same experiences, in the
    same order, as insert_batch / sample_batch -- including the fall-back when the ring wraps"""
    from surreal_amd.replay import FIFOReplay
    from surreal_amd.env import SyntheticVecEnv
    n, D, A, T = 4, 5, 2, 6
    lc, ec, sc = configs(B=4, N=T, stride=T, D=D, A=A, memory=9)          # capacity 12
    a, b = FIFOReplay(lc, ec, sc), FIFOReplay(lc, ec, sc)
    venv = SyntheticVecEnv(n, D, A, episode_len=T, seeds=[1, 2, 3, 4])
    g = torch.Generator().manual_seed(0)
    took_view, took_copy = 0, 0
    for it in range(7):
        venv.reset()
        venv.start_rollout(T, info_width=2 * A)
        for t in range(T):
            venv.step(torch.randn(n, A, generator=g), pds=torch.full((n, 2 * A), float(10 * it + t)))
        ref = venv.emit_windows(T, T)
        a.insert_batch(ref)
        slots = b.reserve_batch(n, venv.window_shapes(T))
        if slots is not None:
            venv.emit_windows(T, T, out=slots)
            b.commit_batch(n)
            took_view += 1
        else:
            b.insert_batch(venv.emit_windows(T, T))
            took_copy += 1
        assert len(a) == len(b)
        if it % 2 == 1 or it == 6:
            pa, pb = a.sample_batch(4), b.sample_batch(4, copy=False)
            assert set(pa) == set(pb)
            for k in pa:
                assert pa[k].shape == pb[k].shape and torch.equal(pa[k], pb[k]), k
    assert took_view >= 3 and took_copy >= 1 and a.cumulative_collected_count == b.cumulative_collected_count


def _rollout_pair(K_name=None):
    """(fused rollout, act_batch + step loop) on the same agent / env / noise"""
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticVecEnv
    n, D, A, T = 6, 7, 3, 5
    lc, ec, sc = configs(B=n, N=T, stride=T, D=D, A=A)
    agent = PPOAgent(lc, ec, sc, agent_id=2, agent_mode='training')
    dev = agent.device
    zf = agent.model.z_filter
    mean, var = torch.linspace(-3, 3, D), torch.linspace(0.5, 2, D)
    zf.running_sum.copy_(mean * 40)
    zf.running_sumsq.copy_((mean * mean + var) * 40)
    zf.count.fill_(40.0)
    eps = torch.randn(T, n, A, generator=torch.Generator().manual_seed(4)).to(dev)
    outs = []
    for fused in (True, False):
        venv = SyntheticVecEnv(n, D, A, episode_len=T, seeds=list(range(n)))
        venv.start_rollout(T, info_width=2 * A)
        if fused:
            venv.rollout(agent, eps=eps)
        else:
            for t in range(T):
                a, pd = agent.act_batch(venv.state, eps=eps[t])
                venv.step(a, pds=pd)
        assert venv.slot == T and venv.t == 0
        outs.append({k: v.cpu().clone() for k, v in venv.emit_windows(T, T).items()})
        outs[-1]['state'] = venv.state.cpu().clone()
    return outs


def test_fused_rollout_equals_act_batch_loop(cpu_double):
    """SyntheticVecEnv.rollout (4 launches per step) records exactly what the per-step loop records"""
    fused, loop = _rollout_pair()
    assert set(fused) == set(loop)
    for k in fused:
        assert torch.equal(fused[k], loop[k]), k
    assert float(fused['pds'].abs().sum()) > 0 and float(fused['dones'][:, -1].min()) == 1.0


def test_pipeline_bench_feeds_several_learns_per_rollout(cpu_double):
    """scripts/bench_pipeline.py (bench.py's on-device-loop secondary) on the CPU double: more actors than one learner
    batch -> the FIFO hands the rollout over in learner batches; every stage is exercised and the counters add up"""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    bp = importlib.import_module('bench_pipeline')
    r = bp.run_pipeline(actors=8, steps=6, obs_dim=8, action_dim=2, iters=1, warmup=0, graph=False, fused_step=True,
                        cpu_double=True, learn_batch=4)
    assert r['config']['learns_per_rollout'] == 2 and r['config']['learn_batch'] == 4
    assert r['value'] > 0 and set(r['stage_ms_synchronised']) == {'rollout', 'windows+fifo', 'learn'}


def test_parameter_noise_matches_reference():
    """agent/param_noise.py:9-74 on the wire form of fetched parameters: same numpy draws, same
    sigma adaptation (incl. the reference's distance bookkeeping) as the golden recorded from it"""
    from surreal_amd.agent.param_noise import NormalParameterNoise, AdaptiveNormalParameterNoise
    C = collections

    def params0():
        return C.OrderedDict(ddpg=C.OrderedDict([('actor.w', np.arange(6, dtype=np.float32).reshape(2, 3)),
                                                 ('actor.b', np.array([0.5, -0.5], dtype=np.float32)),
                                                 ('critic.w', np.linspace(-1, 1, 4).astype(np.float32))]))
    np.random.seed(11)
    got = NormalParameterNoise(0.25).apply(params0())
    for k, v in GOLD['param_noise_normal_seed11'].items():
        np.testing.assert_array_equal(np.asarray(got['ddpg'][k]), np.asarray(v))

    class CleanModel(object):
        loaded = None

        def __call__(self, obs, calculate_value=False):
            return np.asarray(obs, dtype=np.float64) * float(self.loaded['ddpg']['actor.b'][0]), None

    class Loader(object):
        def __init__(self, m):
            self.m = m

        def load(self, params):
            self.m.loaded = params
    clean = CleanModel()
    an = AdaptiveNormalParameterNoise(clean, Loader(clean), target_stddev=0.25, compute_dist_interval=3,
                                      alpha=1.5, sigma=0.1)
    np.random.seed(12)
    p = params0()
    for rnd, want in enumerate(GOLD['param_noise_adaptive_seed12']):
        p = an.apply(p)
        assert an.sigma == want['sigma'] and float(p['ddpg']['actor.b'][0]) == want['b0']
        assert float(clean.loaded['ddpg']['actor.b'][0]) == want['clean_b0']
        for t in range(5 + rnd):
            an.compute_action_distance(np.array([1.0, 2.0]), np.array([0.1 * (t + 1) * (rnd + 1), 0.0]))
        assert an.i == want['i'] and float(an.total_action_distance) == want['dist']
    with pytest.raises(AssertionError):
        NormalParameterNoise(0.1).apply({'m': {'k': [1.0, 2.0]}})


def test_ddpg_agent_parameter_noise_in_the_fetch_path(cpu_double):
    """the agent perturbs what it fetches (both hand-offs) and, in adaptive mode, keeps a clean copy of
    the policy whose distance to the noisy one drives sigma"""
    from surreal_amd.agent import DDPGAgent
    from surreal_amd.learner import DDPGLearner
    from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config
    D, A = 6, 2
    lc = ddpg_learner_config()
    lc.model.actor_fc_hidden_sizes, lc.model.critic_fc_hidden_sizes = [12, 8], [16, 12]
    lc.algo.exploration.param_noise_type = 'adaptive_normal'
    lc.algo.exploration.param_noise_sigma = 0.05
    lc.replay.batch_size = 4
    ec, sc = ddpg_env_config(D, A), ddpg_session_config()
    learner = DDPGLearner(lc, ec, sc)
    ag = DDPGAgent(lc, ec, sc, agent_id=1, agent_mode='training')
    ag.attach_learner(learner)
    np.random.seed(5)
    assert ag.fetch_parameter()
    clean = ag.param_noise.original_model
    assert torch.equal(clean.actor_flat, learner.model.actor_flat)            # the clean copy
    d = (ag.model.actor_flat - learner.model.actor_flat)
    assert 0.03 < float(d.std()) < 0.07 and not torch.equal(ag.model.critic_flat, learner.model.critic_flat)
    obs = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=np.linspace(-1, 1, D).astype(np.float32)))
    for _ in range(3):
        a = ag.act(obs)
        assert a.shape == (A,) and np.all(np.abs(a) <= 1)
    assert ag.param_noise.i == 3 and ag.param_noise.total_action_distance > 0
    sigma0 = ag.param_noise.sigma
    learner._publish_for_test = True
    learner.publish_parameter(1, message='t')
    assert ag.fetch_parameter() and ag.param_noise.sigma != sigma0 and ag.param_noise.i == 0
    plain = DDPGAgent(ddpg_learner_config(), ec, sc, agent_id=1, agent_mode='training')
    assert plain.param_noise is None


def test_device_camera_and_frame_stacking_match_host_wrapper(cpu_double):
    """SyntheticVecEnv(pixel, frame_stacks) on the torch-CPU double == SyntheticEnv under FrameStackWrapper"""
    import pixel_env_cases as PC
    PC.check_device_camera_matches_host_framestack()
    from surreal_amd import kernels as KN
    PC.check_frame_stack_with_resets_inside(KN.default_kernels(), 'cpu')


def test_host_fed_learner_through_the_prefetcher(cpu_double):
    """LearnerDataPrefetcher + PinnedBatchStager (host tier: plain memory, no copy stream): in-place aggregation into
    the staging slots, slot hand-over, same results as synchronous feeding"""
    import wire_cases
    wire_cases.check_host_fed_learner(expect_cuda=False)


def test_aggregation_pool_fills_disjoint_row_ranges_of_one_shared_slot():
    """three worker PROCESSES aggregate their row ranges of a batch in place, into shared-memory staging
    (surreal_amd.distributed.AggregationPool; the reference's prefetch_processes, data_fetcher.py:36-45): the slot
    equals the single-process aggregate of the same experiences, for two consecutive batches on two slots"""
    import functools
    from surreal_amd import synthetic
    from surreal_amd.distributed import SharedBatchStager, AggregationPool, PooledDataPrefetcher, ppo_aggregate_factory
    from surreal_amd.main.ppo_configs import ppo_env_config, ppo_session_config
    from surreal_amd.learner.aggregator import MultistepAggregatorWithInfo
    B, N, D, A = 13, 6, 5, 3
    ec = ppo_env_config(D, A)
    agg = MultistepAggregatorWithInfo(ec.obs_spec, ec.action_spec)
    want = [agg.aggregate(synthetic.ppo_experiences(synthetic.make_ppo_batch(B, N, D, A, seed=40 + k))) for k in range(2)]
    stager = SharedBatchStager(want[0], depth=2, device='cpu')
    pool = AggregationPool(stager, 3, synthetic.SyntheticExperienceSource(B, N, D, A, seed0=40, distinct=2, fresh=True),
                           functools.partial(ppo_aggregate_factory, dict(ec.obs_spec), dict(ec.action_spec)))
    try:
        pf = PooledDataPrefetcher(ppo_session_config('/tmp/smx_pool_test'), B, pool)
        pf.start()
        for k in range(4):
            got = pf.get()
            w = want[k % 2]
            np.testing.assert_array_equal(got['obs']['low_dim']['flat_inputs'].numpy(), w['obs']['low_dim']['flat_inputs'])
            np.testing.assert_array_equal(got['obs_next']['low_dim']['flat_inputs'].numpy(), w['obs_next']['low_dim']['flat_inputs'])
            for name in ('actions', 'rewards', 'dones'):
                np.testing.assert_array_equal(got[name].numpy(), np.asarray(w[name], dtype=np.float32))
            np.testing.assert_array_equal(got['persistent_infos'][0].numpy(), w['persistent_infos'][0])
        pf.stop()
    finally:
        pool.close()
        stager.close()


def test_pooled_host_fed_learner_equals_synchronous_feed(cpu_double):
    """the learner fed by worker processes through the shared staging slot (host tier) -- same results as handing it
    the batches synchronously"""
    import wire_cases
    wire_cases.check_pooled_host_fed_learner(expect_cuda=False, workers=2)
