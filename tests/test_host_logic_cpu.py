"""CPU tier: the product PPOLearner's HOST logic (epoch control, device-side early-exit
protocol, statistics slots, forward reuse, Adam step bookkeeping) driven through the torch-CPU
kernel test double must reproduce the reference's golden traces.  This isolates orchestration
bugs from kernel bugs; the kernels themselves are checked on the GPU (-m gpu)."""
import copy

import numpy as np
import pytest

import helpers as H

ALL_CASES = H.golden_cases(big=False)        # MLP policy and LSTM-stem policy (the reference default)


@pytest.mark.parametrize('name', ALL_CASES)
def test_learner_host_logic_matches_reference(name, cpu_double):
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate)
    stats = learner.learn(copy.deepcopy(batch))
    ws = learner._ws
    np.testing.assert_allclose(ws.adv.numpy().reshape(g['advantages'].shape), g['advantages'],
                               atol=H.ATOL, rtol=H.RTOL)
    np.testing.assert_allclose(ws.ret.numpy().reshape(g['returns'].shape), g['returns'],
                               atol=H.ATOL, rtol=H.RTOL)
    H.assert_trace_close(learner.trace, g, what=name)
    H.assert_stats_close(stats, g, what=name)
    H.assert_final_params(learner, g, case, what=name)
    if zstate is not None:
        sd = learner.model.z_filter.state_dict()
        for k in ('running_sum', 'running_sumsq', 'count'):
            np.testing.assert_allclose(sd[k].numpy(), g['zfinal.' + k], rtol=1e-6)


def test_two_stream_schedule_host_logic(cpu_double):
    g, case = H.load_golden('tiny_adapt_cutoff2')
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate, session_overrides={'epoch_schedule': 'two_stream'})
    stats = learner.learn(copy.deepcopy(batch))
    H.assert_trace_close(learner.trace, g, what='two_stream')
    H.assert_stats_close(stats, g, what='two_stream')


def test_second_learn_continues_adam_state(cpu_double):
    """Adam moments and step counters persist across learn() calls, as torch.optim does"""
    import ppo_oracle
    g, case = H.load_golden('tiny_clip')
    batch, params, zstate = H.case_inputs(case)
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate,
                                    **hyper)
    learner = H.make_learner(case, params, zstate)
    for it in range(3):
        b = dict(batch)
        so = O.learn(copy.deepcopy(b))
        sl = learner.learn(copy.deepcopy(b))
        for k in so:
            if k != '_lr':
                np.testing.assert_allclose(sl[k], so[k], atol=H.ATOL, rtol=H.RTOL,
                                           err_msg='iter %d %s' % (it, k))


def test_post_publish_adapts_coefficients(cpu_double):
    """beta / clip-epsilon adaptation and the reference-policy refresh (ppo.py:637-666)"""
    g, case = H.load_golden('tiny_adapt')
    batch, params, zstate = H.case_inputs(case)
    L = H.make_learner(case, params, zstate)
    L.learn(copy.deepcopy(batch))
    L.kl_record = [L.kl_target * 3.0]
    L._post_publish()
    assert L.beta == pytest.approx(1.5) and L.kl_record == [] and L.exp_counter == 0
    L.kl_record = [L.kl_target * 0.1]
    L._post_publish()
    assert L.beta == pytest.approx(1.0)
    np.testing.assert_array_equal(L.ref_target_model.actor_flat.numpy(), L.model.actor_flat.numpy())
    case2 = copy.deepcopy(case)
    case2['hyper']['ppo_mode'] = 'clip'
    L = H.make_learner(case2, params, zstate)
    L.kl_record = [1.0]
    L._post_publish()
    assert L.clip_epsilon == pytest.approx(0.2 / 1.2)
    L.kl_record = [0.0]
    L._post_publish()
    L.kl_record = [0.0]
    L._post_publish()
    assert L.clip_epsilon == pytest.approx(0.2 * 1.2)


@pytest.mark.parametrize('name', ['tiny_pixel_clip', 'tiny_pixel_rnn_adapt'])
def test_cnn_stem_chunked_critic_pass_with_a_tail(name, cpu_double):
    """the critic pass runs the CNN stem in chunks of cnn_chunk_frames; a chunk size that does not
    divide B*(N+1) leaves a shorter tail chunk inside the same workspace"""
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    frames = case['shape']['B'] * (case['shape']['N'] + 1)
    chunk = max(2, frames // 3 + 1)
    assert frames % chunk != 0
    learner = H.make_learner(case, params, zstate, session_overrides={'cnn_chunk_frames': chunk})
    stats = learner.learn(copy.deepcopy(batch))
    H.assert_trace_close(learner.trace, g, what=name)
    H.assert_stats_close(stats, g, what=name)


def test_deferred_stats_and_scalar_recorder():
    """DeferredStats resolves once, when first looked at; ScalarRecorder stores it untouched and folds
    it into `latest` on demand (also when old history is trimmed); it pickles as a plain dict"""
    import pickle
    from surreal_amd.learner.base import DeferredStats, ScalarRecorder
    calls = []

    def make(value):
        box = {}

        def resolve():
            calls.append(value['k'])
            box['d']._value = dict(value)
        box['d'] = DeferredStats(resolve)
        return box['d']
    rec = ScalarRecorder(keep=3)
    a, b = make({'k': 1, 'x': 1.0}), make({'k': 2, 'y': 2.0})
    rec.add_scalars(a, 1)
    rec.add_scalars({'k': 0, 'z': 9.0}, 2)
    rec.add_scalars(b, 3)
    assert calls == []                                   # nothing has been looked at
    assert 'x' in a and calls == [1] and a['x'] == 1.0 and len(a) == 2 and calls == [1]
    assert rec.latest == {'k': 2, 'x': 1.0, 'z': 9.0, 'y': 2.0} and calls == [1, 2]
    c = make({'k': 3, 'w': 4.0})
    rec.add_scalars(c, 4)
    rec.add_scalars({'k': 5}, 5)                         # trims: everything dropped was folded first
    assert len(rec.history) == 3 and rec.latest['w'] == 4.0 and rec.latest['k'] == 5 and rec.latest['x'] == 1.0
    assert pickle.loads(pickle.dumps(c)) == {'k': 3, 'w': 4.0} and c.copy() == dict(c)
    assert 'in flight' in repr(make({'k': 7}))
