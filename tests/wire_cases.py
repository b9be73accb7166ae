"""Shared bodies of the wire-format parity tests that run on both tiers (tests only): the CPU tier on the
torch-CPU kernel double, the GPU tier (-m gpu) on the HIP kernels through the C ABI."""
import copy
import os
import pickle

import numpy as np
import torch

import helpers as H
import ppo_oracle
from surreal_amd import synthetic
from surreal_amd.distributed import ExperienceCollector
from surreal_amd.utils import serializer as S


def check_reference_chunk_to_learn(expect_cuda):
    """f2 against the oracle: the byte chunk the REFERENCE's ExpBuffer produced (exp_sender.py:10-98) ->
    ExperienceCollector (exp_collector.py:37-65) -> FIFOReplay -> MultistepAggregatorWithInfo -> PPOLearner on
    the GPU.  (1) what reaches the device equals, bit for bit, the batch the reference's own collector +
    aggregator made of the same chunk (recorded: exp_chunk_aggregated.npz); (2) learn() on it equals the
    oracle's learn() on the reference's batch at 1e-5."""
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    wire = os.path.join(H.GOLDEN_DIR, 'wire')
    fx = pickle.load(open(os.path.join(wire, 'exp_chunk.pkl'), 'rb'))
    ref = np.load(os.path.join(wire, 'exp_chunk_aggregated.npz'))
    B, N, D, A = ref['obs'].shape[0], ref['obs'].shape[1], ref['obs'].shape[2], ref['actions'].shape[2]
    lc = ppo_learner_config()
    lc.algo.n_step = N
    lc.algo.rnn.if_rnn_policy = False
    lc.replay.batch_size, lc.replay.memory_size, lc.replay.sampling_start_size = B, 16, B
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = [24, 16]
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_gpu_wire_f2')
    S.set_global_serializer(pickle.dumps, pickle.loads)       # what the fixture was written with (see its generator)
    replay = FIFOReplay(lc, ec, sc)
    ExperienceCollector(replay._insert_wrapper).recv(fx['chunk'])
    assert len(replay) == B
    learner = PPOLearner(lc, ec, sc)
    params = synthetic.make_ppo_params(D, A, hidden=(24, 16), seed=31, final_scale=2.0, log_sig_spread=0.3)
    zstate = synthetic.make_zfilter_state(D, seed=9)
    for m in (learner.model, learner.ref_target_model):
        m.load_params(params)
        m.z_filter.load_state_dict(zstate)
    learner.attach_replay(replay)
    data = learner.fetch_batch()                               # sample -> aggregate
    dev = learner._preprocess_batch_ppo(copy.deepcopy(data))   # -> device tensors, as learn() does
    got = {'obs': dev['obs']['low_dim']['flat_inputs'], 'obs_next': dev['obs_next']['low_dim']['flat_inputs'],
           'actions': dev['actions'], 'rewards': dev['rewards'], 'dones': dev['dones'],
           'persistent_infos0': dev['persistent_infos'][0]}
    for k, v in got.items():
        assert v.is_cuda == expect_cuda and v.dtype == torch.float32, k
        np.testing.assert_array_equal(v.cpu().numpy(), ref[k].astype(np.float32), err_msg=k)
    assert dev.get('onetime_infos') is None
    stats = learner.learn(data)
    # the oracle on the REFERENCE's batch
    O = ppo_oracle.OraclePPOLearner(params, A, B, zstate=zstate, n_step=N)
    obs = lambda x: {'low_dim': {'flat_inputs': x}}  # noqa: E731
    want = O.learn({'obs': obs(ref['obs']), 'obs_next': obs(ref['obs_next']), 'actions': ref['actions'],
                    'rewards': ref['rewards'], 'dones': ref['dones'], 'persistent_infos': [ref['persistent_infos0']],
                    'onetime_infos': None})
    assert learner.epochs_executed == len(O.trace['policy'])
    for k, v in want.items():
        if k == '_lr':
            continue
        at, rt = H.tol_for(k, H.ATOL, H.RTOL)
        np.testing.assert_allclose(stats[k], v, atol=at, rtol=rt, err_msg=k)
    for a, b in zip(learner.trace['policy'] + learner.trace['value'], O.trace['policy'] + O.trace['value']):
        for k in b:
            at, rt = H.tol_for(k, H.ATOL, H.RTOL)
            np.testing.assert_allclose(a[k], b[k], atol=at, rtol=rt, err_msg=k)
