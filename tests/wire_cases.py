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


def check_host_fed_learner(expect_cuda):
    """Experiences in HOST memory (a host-tier FIFO filled by the collector) -> LearnerDataPrefetcher: a background
    thread aggregates the next batch straight into pinned struct-of-arrays staging, its host-to-device copy runs on a
    second stream under the current learn().  The learner that is fed this way ends with exactly the parameters and
    statistics of one that is handed the same batches synchronously (surreal/distributed/data_fetcher.py:9-73,
    surreal/learner/base.py:102-110,149-154)."""
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    B, N, D, A, iters = 8, 6, 9, 3, 5
    lc = ppo_learner_config()
    lc.algo.n_step = N
    lc.algo.rnn.if_rnn_policy = False
    lc.replay.batch_size, lc.replay.memory_size, lc.replay.sampling_start_size = B, B * iters, B
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = [24, 16]
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_host_fed')
    params = synthetic.make_ppo_params(D, A, hidden=(24, 16), seed=3, final_scale=2.0, log_sig_spread=0.3)
    zstate = synthetic.make_zfilter_state(D, seed=4)
    rs = np.random.RandomState(8)

    def experience():
        ob = lambda: {'low_dim': {'flat_inputs': rs.randn(D).astype(np.float32)}}  # noqa: E731
        return {'obs': [ob() for _ in range(N)], 'obs_next': ob(), 'actions': [rs.randn(A).astype(np.float32) for _ in range(N)],
                'rewards': [float(rs.randn()) for _ in range(N)], 'dones': [False] * (N - 1) + [bool(rs.rand() < 0.3)],
                'persistent_infos': [[np.concatenate([np.tanh(rs.randn(A)), np.exp(rs.randn(A) * 0.1 - 1)]).astype(np.float32)]
                                     for _ in range(N)], 'onetime_infos': [], 'n_step': N}
    exps = [experience() for _ in range(B * iters)]

    def make():
        replay = FIFOReplay(lc, ec, sc)
        for e in exps:
            replay.insert(e)
        learner = PPOLearner(lc, ec, sc)
        for m in (learner.model, learner.ref_target_model):
            m.load_params(params)
            m.z_filter.load_state_dict(zstate)
        left = [iters]

        def source():                      # (a source that ends: the prefetch thread must not spin on an empty replay)
            if left[0] == 0:
                import time
                time.sleep(3600)
            left[0] -= 1
            return replay.sample(B)
        learner.set_data_source(source)
        return learner
    plain, fed = make(), make()
    want = [dict(plain.learn(plain.fetch_batch())) for _ in range(iters)]
    pf = fed.start_prefetching(depth=2)
    assert pf.stager.bytes_per_batch == 4 * (B * N * D + B * D + B * N * A + 2 * B * N + B * N * 2 * A)
    got, ptrs = [], set()
    for _ in range(iters):
        batch = fed.fetch_batch()
        x = batch['obs']['low_dim']['flat_inputs']
        assert torch.is_tensor(x) and x.is_cuda == expect_cuda
        ptrs.add(x.data_ptr())
        got.append(dict(fed.learn(batch)))
    assert len(ptrs) == 2                   # two staging slots, stable addresses
    if expect_cuda:
        assert len(fed._graphs) == 2        # one captured graph per slot, no per-batch staging copy inside learn()
    for i, (a, b) in enumerate(zip(got, want)):
        for k in b:
            assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), 'learn %d: %s %r vs %r' % (i, k, a[k], b[k])
    assert torch.equal(fed.model.flat, plain.model.flat)
    pf.stop()


def check_preprocess_override_same_batches(expect_cuda):
    """a Learner subclass whose preprocess() does more than the device move (here: halves the rewards and adds a key)
    sees the same batches with and without prefetching (reference: LearnerDataPrefetcher(main_preprocess=
    self.preprocess), surreal/learner/base.py:102-110)"""
    from surreal_amd.learner import PPOLearner
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    B, N, D, A = 6, 5, 7, 2
    lc = ppo_learner_config()
    lc.algo.n_step = N
    lc.algo.rnn.if_rnn_policy = False
    lc.replay.batch_size = B
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = [16, 8]
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_override')
    calls = []

    class Scaled(PPOLearner):
        def preprocess(self, batch):
            calls.append(type(batch['rewards']).__name__)
            r = batch['rewards']
            batch['rewards'] = (r if torch.is_tensor(r) else torch.as_tensor(np.asarray(r), dtype=torch.float32)) * 0.5
            batch['scaled'] = True
            return batch
    batches = [synthetic.make_ppo_batch(B, N, D, A, seed=70 + k) for k in range(3)]

    def make():
        L = Scaled(lc, ec, sc)
        it = iter([L.aggregator.aggregate(synthetic.ppo_experiences(b)) for b in batches])

        def source():
            try:
                return next(it)
            except StopIteration:
                import time
                time.sleep(3600)
        L._prefetcher_preprocess = lambda data: data          # (the source already hands out aggregated batches)
        L.set_data_source(source)
        return L
    plain, fed = make(), make()
    want = [plain.fetch_batch() for _ in range(3)]
    pf = fed.start_prefetching(depth=2)
    assert fed._prefetch_main_preprocess is not None
    for w in want:
        got = fed.fetch_batch()
        assert got['scaled'] is True and w['scaled'] is True
        g, r = got['rewards'], w['rewards']
        assert torch.is_tensor(g) and g.is_cuda == expect_cuda
        np.testing.assert_array_equal(g.cpu().numpy().reshape(-1), np.asarray(r.cpu() if torch.is_tensor(r) else r).reshape(-1))
        np.testing.assert_array_equal(got['obs']['low_dim']['flat_inputs'].cpu().numpy(),
                                      np.asarray(w['obs']['low_dim']['flat_inputs']))
    pf.stop()
    # the hooks the package ships only move the batch: they are skipped under prefetching (the stager did the move)
    base = PPOLearner(lc, ec, sc)
    assert getattr(type(base).preprocess, 'device_move_only', False)


def check_pooled_host_fed_learner(expect_cuda, workers=3):
    """the aggregation in WORKER PROCESSES (surreal_amd.distributed.AggregationPool: the reference's prefetch_processes,
    surreal/distributed/data_fetcher.py:36-45): every worker fills its rows of the shared staging slot in place, the
    learner consumes the device twins -- and ends with exactly the statistics and parameters of a learner handed the
    same batches synchronously"""
    import functools
    from surreal_amd.learner import PPOLearner
    from surreal_amd.distributed import SharedBatchStager, AggregationPool, PooledDataPrefetcher, ppo_aggregate_factory
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    B, N, D, A, iters = 22, 6, 9, 3, 5
    lc = ppo_learner_config()
    lc.algo.n_step = N
    lc.algo.rnn.if_rnn_policy = False
    lc.replay.batch_size = B
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = [24, 16]
    ec, sc = ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_test_pooled')
    params = synthetic.make_ppo_params(D, A, hidden=(24, 16), seed=3, final_scale=2.0, log_sig_spread=0.3)
    zstate = synthetic.make_zfilter_state(D, seed=4)

    def make():
        learner = PPOLearner(lc, ec, sc)
        for m in (learner.model, learner.ref_target_model):
            m.load_params(params)
            m.z_filter.load_state_dict(zstate)
        return learner
    plain, fed = make(), make()
    batches = [synthetic.make_ppo_batch(B, N, D, A, seed=60 + k) for k in range(3)]
    want = []
    for i in range(iters):
        b = plain.aggregator.aggregate(synthetic.ppo_experiences(batches[i % 3]))
        want.append(dict(plain.learn(plain._preprocess_batch_ppo(b))))
    example = fed.aggregator.aggregate(synthetic.ppo_experiences(batches[0]))
    fed.graph_input_sets = 2
    stager = SharedBatchStager(example, depth=2, device=fed.device)
    pool = AggregationPool(stager, workers, synthetic.SyntheticExperienceSource(B, N, D, A, seed0=60, distinct=3, fresh=True),
                           functools.partial(ppo_aggregate_factory, ec.obs_spec.to_dict(), ec.action_spec.to_dict()))
    try:
        pf = PooledDataPrefetcher(sc, B, pool)
        pf.start()
        got = []
        for _ in range(iters):
            batch = pf.get()
            x = batch['obs']['low_dim']['flat_inputs']
            assert torch.is_tensor(x) and x.is_cuda == expect_cuda
            got.append(dict(fed.learn(batch)))
        for i, (a, b) in enumerate(zip(got, want)):
            for k in b:
                assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), 'learn %d: %s %r vs %r' % (i, k, a[k], b[k])
        assert torch.equal(fed.model.flat, plain.model.flat)
        pf.stop()
    finally:
        pool.close()
        if expect_cuda:
            torch.cuda.synchronize()
        stager.close()
