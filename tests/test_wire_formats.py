"""CPU tier: the SURVEY.md section 8(f) formats -- experience chunks, the parameter protocol,
parameter blobs, checkpoints -- against fixtures recorded from the reference's own code
(tests/golden/wire/, made by oracle/gen_golden_wire.py, which also proves the opposite direction:
the reference consumes what surreal_amd writes)."""
import collections
import copy
import json
import os
import pickle
import shutil

import numpy as np
import pytest
import torch
import yaml

import helpers as H
from surreal_amd.distributed import (ExpBuffer, ExpSender, ExperienceCollector, ModuleDict,
                                     ParameterClient, ParameterPublisher, ParameterServer)
from surreal_amd.utils import serializer as S
from surreal_amd.utils.checkpoint import Checkpoint, PeriodicCheckpoint

WIRE = os.path.join(H.GOLDEN_DIR, 'wire')


def make_experiences(n=5, n_step=3, D=4, A=2, seed=0):
    """same generator as oracle/gen_golden_wire.py"""
    rs = np.random.RandomState(seed)
    obs_seq = [collections.OrderedDict(low_dim=collections.OrderedDict(
        flat_inputs=rs.randn(D).astype(np.float32))) for _ in range(n + n_step)]
    exps = []
    for i in range(n):
        exps.append({
            'obs': [obs_seq[i + k] for k in range(n_step)],
            'obs_next': obs_seq[i + n_step],
            'actions': [rs.randn(A).astype(np.float32) for _ in range(n_step)],
            'onetime_infos': [],
            'persistent_infos': [[rs.randn(2 * A).astype(np.float32)] for _ in range(n_step)],
            'rewards': [float(rs.randn()) for _ in range(n_step)],
            'dones': [False] * (n_step - 1) + [bool(i == n - 1)],
            'infos': [{} for _ in range(n_step)],
            'n_step': n_step,
        })
    return exps


def same(a, b):
    if isinstance(a, dict):
        assert isinstance(b, dict) and sorted(a.keys()) == sorted(b.keys())     # order is not part of the contract
        for k in a:
            same(a[k], b[k])
    elif isinstance(a, (list, tuple)):
        assert type(a) is type(b) and len(a) == len(b)
        for x, y in zip(a, b):
            same(x, y)
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    else:
        assert a == b


def test_experience_chunk_matches_reference():
    fx = pickle.load(open(os.path.join(WIRE, 'exp_chunk.pkl'), 'rb'))
    exps = make_experiences()
    # agent side: same chunk as the reference's ExpBuffer, byte for byte; 8 distinct observations
    # travel once although 5 windows x 4 slots reference them
    sent = []
    sender = ExpSender(send_fn=sent.append, flush_iteration=len(exps))
    hashes = [sender.send_exp(e) for e in exps]
    assert hashes[:-1] == [None] * (len(exps) - 1) and hashes[-1] == fx['chunk_hash']
    assert sent == [fx['chunk']]
    exp_list, storage = S.deserialize(sent[0])
    assert len(storage) == fx['n_storage'] == 8
    assert 'obs_hash' in exp_list[0] and 'obs' not in exp_list[0] and exp_list[0]['n_step'] == 3
    # replay side: the reference's chunk unpacks to what the reference's collector makes of it
    got = []
    n = ExperienceCollector(got.append).recv(fx['chunk'])
    assert n == len(exps)
    same(got, fx['unpacked'])
    same(got, exps)
    # overlapping windows share the observation OBJECTS after unpacking, as in the reference
    assert got[0]['obs'][1]['low_dim']['flat_inputs'] is got[1]['obs'][0]['low_dim']['flat_inputs']


def test_exp_buffer_argument_checks():
    b = ExpBuffer()
    with pytest.raises(AssertionError):
        b.add({'obs_hash': 1}, {})
    with pytest.raises(TypeError):
        b.add([1], {})
    assert b._hash_nested(None) is None and b._hash_nested((None, [None])) == (None, [None])


def test_parameter_protocol_matches_reference():
    fx = json.load(open(os.path.join(WIRE, 'param_protocol.json')))
    assert S.binary_hash(b'') == fx['hashes'][''] and S.binary_hash(b'abc') == fx['hashes']['abc']
    assert S.string_hash('surreal') == fx['hashes']['string_hash(surreal)']
    ps = ParameterServer()
    assert list(ps.handle_request('parameter')) == fx['empty_parameter']
    assert list(ps.handle_request('info')) == fx['empty_info']
    binary = bytes.fromhex(fx['binary_hex'])
    info = dict(fx['info'][1])
    assert info['hash'] == S.binary_hash(binary)
    ps.set_storage((binary, info))
    for req, key in (('info', 'info'), ('parameter', 'parameter'),
                     ('parameter:' + info['hash'], 'parameter_same_hash'),
                     ('parameter:abc', 'parameter_other_hash'), ('parameter:', 'parameter_empty_hash')):
        p, i = ps.handle_request(req)
        assert [None if p is None else p.hex(), i] == fx[key], req
    with pytest.raises(ValueError):
        ps.handle_request('bogus')
    # client-side caching (parameter_server.py:243-271)
    c = ParameterClient(ps.handle_request)
    p, i = c.fetch_parameter_with_info()
    assert p == binary and i['iteration'] == 7 and c.alive
    assert c.fetch_parameter_with_info() == (None, info)             # unchanged -> no download
    assert c.fetch_parameter_with_info(force_update=True)[0] == binary
    assert c.fetch_info() == info

    def down(_):
        raise TimeoutError()
    dead = ParameterClient(down)
    assert dead.fetch_parameter_with_info() == (None, None) and dead.fetch_info() is None and not dead.alive


class Stub(object):
    def __init__(self):
        self.sd = collections.OrderedDict()

    def state_dict(self):
        return self.sd

    def load_state_dict(self, sd):
        self.sd = collections.OrderedDict(sd)


def test_module_dict_blob_matches_reference():
    blob = open(os.path.join(WIRE, 'module_dict.pkl'), 'rb').read()
    ref = S.deserialize(blob)
    assert list(ref.keys()) == ['net'] and list(ref['net'].keys()) == ['fc.weight', 'fc.bias']
    stub = Stub()
    ModuleDict({'net': stub}).loads(blob)
    assert stub.sd['fc.weight'].dtype == np.float32
    np.testing.assert_array_equal(stub.sd['fc.weight'], np.arange(6, dtype=np.float32).reshape(2, 3) / 10)
    # dumps -> the same structure (numpy float32 leaves), and key_map renames at the boundary
    stub.sd = collections.OrderedDict((k, torch.as_tensor(v)) for k, v in stub.sd.items())
    assert S.deserialize(ModuleDict({'net': stub}).dumps())['net']['fc.bias'].tolist() == [0.5, -0.5]
    mapped = S.deserialize(ModuleDict({'net': stub}, key_map={'fc.weight': 'model.0.weight'}).dumps())
    assert list(mapped['net'].keys()) == ['model.0.weight', 'fc.bias']
    with pytest.raises(TypeError):
        ModuleDict({'net': object()})


def test_learner_publishes_through_the_parameter_server(cpu_double):
    """learner.module_dict() -> ParameterPublisher -> ParameterServer -> ParameterClient -> agent model"""
    g, case = H.load_golden('tiny_rnn_clip')
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate)
    other = H.make_learner(case, {k: v * 0 + 0.25 for k, v in params.items()}, zstate)
    ps = ParameterServer()
    pub = ParameterPublisher(ps.set_storage, learner.module_dict())
    info = pub.publish(iteration=3, message='hello')
    client = ParameterClient(ps.handle_request)
    binary, got_info = client.fetch_parameter_with_info()
    assert got_info == info and info['hash'] == S.binary_hash(binary)
    ModuleDict(other.module_dict()).loads(binary)
    assert torch.equal(other.model.flat, learner.model.flat)
    for k in ('running_sum', 'running_sumsq', 'count'):
        assert torch.equal(getattr(other.model.z_filter, k), getattr(learner.model.z_filter, k))
    assert client.fetch_parameter_with_info()[0] is None             # nothing new
    learner.learn(copy.deepcopy(batch))
    pub.publish(iteration=4)
    assert client.fetch_parameter_with_info()[0] is not None         # parameters changed -> new hash


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(3, 2)


class Tracked(object):
    pass


def test_checkpoint_reads_the_reference_folder(tmp_path):
    folder = str(tmp_path / 'ckpt')
    shutil.copytree(os.path.join(WIRE, 'ckpt'), folder)
    t = Tracked()
    t.model, t.counter = Tiny(), -1
    ck = Checkpoint(folder, 'learner', tracked_obj=t, tracked_attrs=None)
    assert ck.metadata['tracked_attrs'] == ['model', 'counter'] and ck.metadata['global_steps'] == 30
    assert ck.metadata['history_ckpt_files'] == ['learner.30.ckpt', 'learner.20.ckpt']
    assert ck.metadata['best_ckpt_files'] == ['learner.best-20.ckpt'] and ck.metadata['best_scores'] == [3.0]
    assert ck.restore(0, 'history', check_ckpt_exists=True).endswith('learner.30.ckpt')
    assert t.counter == 30 and t.model.fc.bias.tolist() == [30.0, 30.0]
    assert ck.restore(1, 'history', check_ckpt_exists=True) and t.counter == 20
    assert ck.restore(0, 'best', check_ckpt_exists=True) and t.model.fc.bias.tolist() == [20.0, 20.0]
    assert ck.restore('30', 'history') and t.counter == 30
    assert ck.restore(5, 'history') is None
    with pytest.raises(FileNotFoundError):
        ck.restore(5, 'history', check_ckpt_exists=True)


def test_checkpoint_writes_the_reference_layout(tmp_path):
    ref_meta = yaml.safe_load(open(os.path.join(WIRE, 'ckpt', 'metadata.learner.yml')))
    t = Tracked()
    t.model, t.counter = Tiny(), 0
    ck = Checkpoint(str(tmp_path), 'learner', tracked_obj=t, tracked_attrs=['model', 'counter'],
                    keep_history=2, keep_best=1)
    for step, score in ((10, 1.0), (20, 3.0), (30, 2.0)):
        t.counter = step
        with torch.no_grad():
            t.model.fc.bias.fill_(float(step))
        ck.save(score=score, global_steps=step)
    assert sorted(os.listdir(str(tmp_path))) == sorted(os.listdir(os.path.join(WIRE, 'ckpt')))
    meta = yaml.safe_load(open(ck.metadata_path()))
    for k in ('version', 'save_counter', 'global_steps', 'tracked_attrs', 'keep_history', 'keep_best',
              'history_ckpt_files', 'best_ckpt_files', 'best_scores'):
        assert meta[k] == ref_meta[k], k
    assert set(meta['ckpt'].keys()) == set(ref_meta['ckpt'].keys())
    for f, entry in meta['ckpt'].items():
        for k in ('score', 'global_steps', 'save_counter'):
            assert entry[k] == ref_meta['ckpt'][f][k]
    ours = pickle.load(open(ck.ckpt_path(30), 'rb'))
    theirs = pickle.load(open(os.path.join(WIRE, 'ckpt', 'learner.30.ckpt'), 'rb'))
    assert list(ours.keys()) == list(theirs.keys()) == ['model', 'counter']
    assert list(ours['model'].keys()) == list(theirs['model'].keys())
    assert torch.equal(ours['model']['fc.bias'], theirs['model']['fc.bias'])
    # periodic variant
    pc = PeriodicCheckpoint(str(tmp_path / 'p'), 'agent', tracked_obj=t, tracked_attrs=['counter'],
                            keep_history=1, keep_best=0, period=3)
    assert [pc.save(global_steps=i) for i in range(1, 7)] == [False, False, True, False, False, True]


def test_learner_checkpoint_round_trip(cpu_double, tmp_path):
    g, case = H.load_golden('tiny_adapt')
    batch, params, zstate = H.case_inputs(case)
    a = H.make_learner(case, params, zstate, session_overrides=None)
    a.session_config.folder = str(tmp_path)
    a._setup_checkpoint()
    a.learn(copy.deepcopy(batch))
    path = a.save_checkpoint(global_steps=a.current_iteration)
    assert os.path.basename(path) == 'learner.1.ckpt'
    assert os.path.exists(os.path.join(str(tmp_path), 'checkpoint', 'metadata.learner.yml'))
    b = H.make_learner(case, {k: v * 0 for k, v in params.items()}, zstate)
    b.session_config.folder = str(tmp_path)
    b._setup_checkpoint()
    assert b.restore_checkpoint()
    assert torch.equal(a.model.flat, b.model.flat) and b.current_iteration == 1
    assert torch.equal(a.ref_target_model.flat, b.ref_target_model.flat)
    assert b.actor_lr_scheduler.state_dict() == a.actor_lr_scheduler.state_dict()


def test_agent_replay_learner_loop_over_the_wire_formats(cpu_double):
    """the in-process hand-offs replaced by the reference's formats end to end: experience chunks
    (hash-deduplicated observations) agent -> replay, parameter blobs learner -> server -> agent"""
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticEnv
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    D, A, N = 6, 2, 5
    lc = ppo_learner_config()
    lc.algo.n_step, lc.algo.stride = N, 2
    lc.algo.rnn.if_rnn_policy = False
    lc.replay.batch_size, lc.replay.memory_size, lc.replay.sampling_start_size = 3, 16, 3
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = [24, 16]
    lc.parameter_publish.exp_interval = 3
    lc.parameter_publish.min_publish_interval = 0.0
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_wire')
    ec.limit_episode_length = 11
    replay = FIFOReplay(lc, ec, sc)
    learner = PPOLearner(lc, ec, sc)
    learner.attach_replay(replay)
    chunks = []
    collector = ExperienceCollector(replay._insert_wrapper)

    def wire(binary):                       # "the socket"
        chunks.append(len(binary))
        collector.recv(binary)
    sender = ExpSender(send_fn=wire, flush_iteration=sc.sender.flush_iteration)
    ps = ParameterServer()
    learner.attach_parameter_publisher(ps.set_storage)
    ag = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='training')
    ag.set_experience_sink(sender.send_exp)
    ag.set_env_factory(lambda: SyntheticEnv(D, A, episode_len=50, seed=1))
    ag.attach_parameter_client(ParameterClient(ps.handle_request))
    ag.main_setup()
    learner.main_setup()
    ag.main_loop()
    ag.main_loop()
    n_windows = 2 * ((11 - N) // 2 + 1)
    assert len(chunks) == n_windows // sc.sender.flush_iteration
    assert replay.cumulative_collected_count == len(chunks) * sc.sender.flush_iteration
    assert not ag.fetch_parameter()                   # nothing published yet
    learner.main_loop()
    assert ps.param_info is not None and ps.param_info['message'] == 'batch 0'     # published once
    before = ag.model.flat.clone()
    assert ag.fetch_parameter() and not ag.fetch_parameter()
    assert torch.equal(ag.model.flat, learner.model.flat) and not torch.equal(before, ag.model.flat)


def test_reference_chunk_to_learn_equals_reference_batch_and_oracle(cpu_double):
    """f2 end to end on the host tier: reference chunk -> collector -> FIFO -> aggregator -> learn"""
    import wire_cases
    wire_cases.check_reference_chunk_to_learn(expect_cuda=False)


def test_overriding_preprocess_sees_the_same_batches_with_and_without_prefetching(cpu_double):
    import wire_cases
    wire_cases.check_preprocess_override_same_batches(expect_cuda=False)


def test_checkpoint_keeps_dict_subclasses_of_tracked_attributes(tmp_path):
    """a tracked attribute that is a defaultdict / Counter: saved and restored with its type and items"""
    import collections
    t = Tracked()
    t.visits = collections.defaultdict(list)
    t.visits['a'].append(torch.ones(2))
    t.counts = collections.Counter({'x': 3})
    ck = Checkpoint(str(tmp_path), 'agent', tracked_obj=t, tracked_attrs=['visits', 'counts'], keep_history=1, keep_best=0)
    ck.save(global_steps=1)
    blob = pickle.load(open(ck.ckpt_path(1), 'rb'))
    assert isinstance(blob['visits'], collections.defaultdict) and blob['visits'].default_factory is list
    assert torch.equal(blob['visits']['a'][0], torch.ones(2))
    assert isinstance(blob['counts'], collections.Counter) and blob['counts'] == collections.Counter({'x': 3})
