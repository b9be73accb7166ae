"""GPU tier (-m gpu): the adjacent formats driven from the DEVICE-resident
learner -- the flat parameter buffers the HIP kernels update in place:

  f1  learner (HIP path) -> ParameterPublisher -> ParameterServer -> ParameterClient -> a PPOAgent whose
      policy then runs on the GPU: the blob is the reference's wire form (numpy fp32 state dict), the
      hash / "unchanged" protocol holds, the agent acts with exactly the learner's parameters
  f2  experience chunks (hash-deduplicated observations) from agents acting on the GPU into the replay
      and on into learn(); and the chunk RECORDED FROM THE REFERENCE's own ExpBuffer (tests/golden/wire)
      through collector -> FIFO replay -> aggregator -> learn on the GPU: the device batch equals the
      reference aggregator's batch of the same chunk bit for bit, the learn() statistics equal the oracle's
  f3  checkpoint round trip in the reference's folder layout: a learner that learned on the GPU is saved
      and restored into a fresh one; the restored learner's next learn() equals the oracle continuing
      from the same parameters (the reference checkpoints the models and schedulers, NOT the optimiser
      state: ppo.py:668-678 -- so the comparison oracle starts with fresh Adam moments too)
"""
import copy
import os
import pickle

import numpy as np
import pytest
import torch

import helpers as H
import ppo_oracle
from surreal_amd import synthetic
from surreal_amd.distributed import (ExpSender, ExperienceCollector, ModuleDict, ParameterClient, ParameterPublisher,
                                     ParameterServer)
from surreal_amd.utils import serializer as S

pytestmark = pytest.mark.gpu


def _learner(name, zero=False):
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    if zero:
        params = {k: v * 0 for k, v in params.items()}
    return g, case, batch, params, zstate, H.make_learner(case, params, zstate)


@pytest.mark.parametrize('name', ['tiny_adapt', 'tiny_rnn_clip', 'cfg5_clip'])
def test_publish_from_device_buffers_through_the_parameter_server(name):
    g, case, batch, params, zstate, learner = _learner(name)
    assert learner.model.flat.is_cuda
    learner.learn(copy.deepcopy(batch))                       # parameters now differ from the injected ones
    ps = ParameterServer()
    pub = ParameterPublisher(ps.set_storage, learner.module_dict())
    info = pub.publish(iteration=1, message='after one learn')
    client = ParameterClient(ps.handle_request)
    binary, got_info = client.fetch_parameter_with_info()
    assert got_info == info and info['hash'] == S.binary_hash(binary)
    blob = S.deserialize(binary)                              # the reference's wire form: numpy float32 leaves
    assert list(blob.keys()) == ['ppo'] and all(isinstance(v, np.ndarray) and v.dtype == np.float32
                                                for v in blob['ppo'].values())
    got = learner.model.numpy_params()
    for k, v in got.items():
        np.testing.assert_array_equal(blob['ppo'][k].reshape(v.shape), v, err_msg=k)
    assert client.fetch_parameter_with_info()[0] is None      # unchanged -> (None, info)
    # an agent fetches through the protocol and then acts on the GPU with the learner's parameters
    from surreal_amd.agent import PPOAgent
    ag = PPOAgent(learner.learner_config, learner.env_config, learner.session_config, agent_id=0,
                  agent_mode='eval_deterministic_local')
    ag.attach_parameter_client(ParameterClient(ps.handle_request))
    assert ag.fetch_parameter() and ag.model.flat.is_cuda
    assert torch.equal(ag.model.flat, learner.model.flat)
    for k in ('running_sum', 'running_sumsq', 'count'):
        assert torch.equal(getattr(ag.model.z_filter, k), getattr(learner.model.z_filter, k))
    shp = case['shape']
    rs = np.random.RandomState(3)
    obs = {'low_dim': {'flat_inputs': rs.randn(shp['D']).astype(np.float32)}}
    a = ag.act(obs)
    hyper = dict(case['hyper'])
    hyper['n_step'] = shp['N']
    O = ppo_oracle.OraclePPOModel(got, shp['A'], True, {k: getattr(learner.model.z_filter, k).cpu().numpy()
                                                        for k in ('running_sum', 'running_sumsq', 'count')})
    cells = None
    if case['hyper'].get('if_rnn_policy'):
        hid = case['rnn_hidden']
        cells = (torch.zeros(1, 1, hid), torch.zeros(1, 1, hid))
    pd = O.forward_actor({'low_dim': {'flat_inputs': torch.as_tensor(obs['low_dim']['flat_inputs'])[None, None]
                                      if cells is not None else torch.as_tensor(obs['low_dim']['flat_inputs'])[None]}},
                         cells).detach().numpy().reshape(-1)
    np.testing.assert_allclose(a, np.clip(pd[:shp['A']], -1, 1), atol=1e-5, rtol=1e-5)
    # another learn changes the parameters -> a new hash is served
    learner.learn(copy.deepcopy(batch))
    pub.publish(iteration=2)
    assert client.fetch_parameter_with_info()[0] is not None


@pytest.mark.parametrize('name', ['tiny_adapt', 'ragged_clip'])
def test_checkpoint_round_trip_from_device_buffers(name, tmp_path):
    g, case, batch, params, zstate, a = _learner(name)
    a.session_config.folder = str(tmp_path)
    a._setup_checkpoint()
    a.learn(copy.deepcopy(batch))
    path = a.save_checkpoint(global_steps=a.current_iteration)
    assert os.path.basename(path) == 'learner.1.ckpt'
    # the file is the reference's layout: a pickle of {attr: state} with CPU tensors in the state dicts
    data = pickle.load(open(path, 'rb'))
    assert list(data.keys()) == a.checkpoint_attributes()
    assert all((not torch.is_tensor(v)) or v.device.type == 'cpu' for v in data['model'].values())
    _, _, _, _, _, b = _learner(name, zero=True)
    b.session_config.folder = str(tmp_path)
    b._setup_checkpoint()
    assert b.restore_checkpoint() and b.current_iteration == 1
    assert b.model.flat.is_cuda and torch.equal(a.model.flat, b.model.flat)
    assert torch.equal(a.ref_target_model.flat, b.ref_target_model.flat)
    for k in ('running_sum', 'running_sumsq', 'count'):
        assert torch.equal(getattr(a.model.z_filter, k), getattr(b.model.z_filter, k))
    # the restored learner's next step == the oracle continuing from the saved parameters with fresh
    # optimiser state (what a restored reference learner does)
    shp = case['shape']
    hyper = dict(case['hyper'])
    hyper['n_step'] = shp['N']
    zs = {k: getattr(a.model.z_filter, k).cpu().numpy() for k in ('running_sum', 'running_sumsq', 'count')}
    O = ppo_oracle.OraclePPOLearner(a.model.numpy_params(), shp['A'], shp['B'], zstate=zs, **hyper)
    O.ref_target_model.load_from(ppo_oracle.OraclePPOModel(a.ref_target_model.numpy_params(), shp['A'], True, zs))
    nxt = synthetic.make_ppo_batch(shp['B'], shp['N'], shp['D'], shp['A'], seed=77, **{
        k: v for k, v in case['batch_args'].items() if k != 'seed'})
    want = O.learn(copy.deepcopy(nxt))
    got = b.learn(copy.deepcopy(nxt))
    assert b.epochs_executed == len(O.trace['policy'])
    for k, v in want.items():
        if k == '_lr':
            continue
        at, rt = H.tol_for(k, H.ATOL, H.RTOL)
        np.testing.assert_allclose(got[k], v, atol=at, rtol=rt, err_msg=k)


def test_experience_chunks_from_gpu_agents_feed_the_learner():
    """agents whose policy runs on the GPU -> windowing wrapper -> ExpSender chunks (observations sent
    once, experiences carry hashes) -> ExperienceCollector -> FIFO replay -> PPOLearner.learn on the GPU"""
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticEnv
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    D, A, N = 8, 2, 5
    lc = ppo_learner_config()
    lc.algo.n_step, lc.algo.stride = N, 2
    lc.algo.rnn.if_rnn_policy = False
    lc.replay.batch_size, lc.replay.memory_size, lc.replay.sampling_start_size = 4, 16, 4
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = [24, 16]
    lc.parameter_publish.exp_interval = 4
    lc.parameter_publish.min_publish_interval = 0.0
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_gpu_wire')
    ec.limit_episode_length = 11
    replay = FIFOReplay(lc, ec, sc)
    learner = PPOLearner(lc, ec, sc)
    learner.attach_replay(replay)
    sizes = []
    collector = ExperienceCollector(replay._insert_wrapper)
    sender = ExpSender(send_fn=lambda b: (sizes.append(len(b)), collector.recv(b)), flush_iteration=sc.sender.flush_iteration)
    ps = ParameterServer()
    learner.attach_parameter_publisher(ps.set_storage)
    ag = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='training')
    assert ag.model.flat.is_cuda
    ag.set_experience_sink(sender.send_exp)
    ag.set_env_factory(lambda: SyntheticEnv(D, A, episode_len=50, seed=1))
    ag.attach_parameter_client(ParameterClient(ps.handle_request))
    ag.main_setup()
    learner.main_setup()
    ag.main_loop()
    ag.main_loop()
    n_windows = 2 * ((11 - N) // 2 + 1)
    assert len(sizes) == n_windows // sc.sender.flush_iteration and len(replay) == len(sizes) * sc.sender.flush_iteration
    learner.main_loop()
    st = learner.tensorplex.latest
    assert all(np.isfinite(st[k]) for k in ('_surr_loss', '_val_loss', '_pol_kl'))
    assert ag.fetch_parameter() and torch.equal(ag.model.flat, learner.model.flat)


def test_reference_chunk_to_device_learn_equals_reference_batch_and_oracle():
    """f2 against the oracle, on the HIP path (body and what it checks: tests/wire_cases.py)"""
    import wire_cases
    wire_cases.check_reference_chunk_to_learn(expect_cuda=True)


def test_host_fed_learner_pinned_double_buffered_ingest():
    """batches from host memory: aggregated in place into PINNED staging by the prefetch thread, copied to the device
    on a second stream while the previous learn() runs, consumed from two fixed address sets (two captured graphs) --
    bit-identical to feeding the same batches synchronously"""
    import wire_cases
    wire_cases.check_host_fed_learner(expect_cuda=True)


def test_host_fed_learner_aggregation_in_worker_processes():
    """the host tier's aggregation spread over worker processes that fill disjoint row ranges of ONE shared,
    host-registered staging slot (body: tests/wire_cases.py)"""
    import wire_cases
    wire_cases.check_pooled_host_fed_learner(expect_cuda=True)


def test_overriding_preprocess_sees_the_same_batches_with_and_without_prefetching():
    import wire_cases
    wire_cases.check_preprocess_override_same_batches(expect_cuda=True)
