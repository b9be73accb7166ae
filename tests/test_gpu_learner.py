"""GPU tier (-m gpu): the product PPOLearner on the HIP path against the golden vectors recorded
from the reference's own CPU learner (tests/golden/, made by oracle/gen_golden.py): advantages,
returns, raw critic values, every per-epoch loss statistic, the number of policy epochs executed
before the KL early exit, final statistics, updated parameters, z-filter state.
Tolerance 1e-5 (abs + rel), fp32 -- BASELINE.json's bound."""
import copy
import json

import numpy as np
import pytest
import torch

import helpers as H
from surreal_amd import _lib as L

pytestmark = pytest.mark.gpu

NON_RNN = H.golden_cases(rnn=False)
RNN = H.golden_cases(rnn=True)


def run_case(name, session_overrides=None):
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate, session_overrides=session_overrides)
    stats = learner.learn(copy.deepcopy(batch))
    return g, case, learner, stats


def check_case(name, g, case, learner, stats):
    ws = learner._ws
    B, N = case['shape']['B'], case['shape']['N']
    H.assert_adv_ret(ws.adv.cpu().numpy(), ws.ret.cpu().numpy(), g, B)
    vr = learner.raw_values().cpu().numpy()
    np.testing.assert_allclose(vr[:g['values_raw'].shape[0]], g['values_raw'], atol=H.ATOL,
                               rtol=H.RTOL, err_msg='raw critic values')
    H.assert_trace_close(learner.trace, g, what=name)
    H.assert_stats_close(stats, g, what=name)
    H.assert_final_params(learner, g, case, what=name)
    if 'zfinal.count' in g:
        sd = learner.model.z_filter.state_dict()
        for k in ('running_sum', 'running_sumsq', 'count'):
            np.testing.assert_allclose(sd[k].cpu().numpy(), g['zfinal.' + k], rtol=2e-6, atol=1e-3)


@pytest.mark.parametrize('name', NON_RNN)
def test_learner_matches_reference_golden_graph(name):
    """default product configuration: hipGraph replay + value epochs on a side stream"""
    check_case(name, *run_case(name))


@pytest.mark.parametrize('name', RNN)
def test_rnn_learner_matches_reference_golden_graph(name):
    """policies with a shared stem -- LSTM (algo.rnn.if_rnn_policy, the reference default; cfg1 is
    the shape of the reference's own test_ppo_gym --unit-test run), CNN over uint8 camera frames
    (cfg4 = 3x84x84 SawyerLift frames + robot state), and both: back-propagation through the
    stems in both optimiser groups, horizon-H windowed GAE over the (B, N+1) critic sequence"""
    check_case(name, *run_case(name))


@pytest.mark.parametrize('name', RNN)
def test_rnn_learner_matches_reference_golden_eager(name):
    check_case(name, *run_case(name, {'use_hip_graph': False}))


@pytest.mark.parametrize('name', ['tiny_adapt_cutoff2', 'ragged_clip', 'cfg2_adapt', 'cfg5_adapt_earlyexit'])
def test_learner_matches_reference_golden_eager(name):
    """same numbers without graph capture"""
    check_case(name, *run_case(name, {'use_hip_graph': False}))


@pytest.mark.parametrize('name,opts', [
    ('cfg5_clip', {'fused_epochs': False}),                      # the layered (one launch per layer) epoch schedule
    ('cfg2_adapt', {'fused_epochs': False}),
    ('cfg5_adapt', {'fused_fwdbwd': False}),                     # forward and backward of an epoch as two launches
    ('cfg5_adapt_earlyexit', {'fused_fwdbwd': False}),
    ('tiny_adapt_cutoff2', {'fused_epochs': False, 'use_hip_graph': False})])
def test_learner_session_options_keep_the_numbers(name, opts):
    """every schedule option the learner still carries (session_config.learner.*) reproduces the same goldens"""
    check_case(name, *run_case(name, opts))


def test_three_learns_match_oracle_and_graph_replays():
    """consecutive learn() calls on device-resident batches: the captured graph is replayed
    (pointer-stable inputs) and the optimiser state carries over exactly as torch.optim's"""
    import ppo_oracle
    g, case = H.load_golden('cfg2_clip')
    batch, params, zstate = H.case_inputs(case)
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
    learner = H.make_learner(case, params, zstate)
    dbatch = learner._preprocess_batch_ppo(copy.deepcopy(batch))      # device-resident once
    for it in range(3):
        so = O.learn(copy.deepcopy(batch))
        sl = learner.learn(dbatch)
        for k in so:
            if k != '_lr':
                at, rt = H.tol_for(k, H.ATOL, 2e-5)
                np.testing.assert_allclose(sl[k], so[k], atol=at, rtol=rt,
                                           err_msg='iteration %d stat %s' % (it, k))
    assert len(learner._graphs) == 1


@pytest.mark.parametrize('name', ['cfg5_adapt', 'tiny_rnn_clip', 'tiny_pixel_clip'])
def test_learns_are_bit_reproducible_run_to_run(name):
    """no atomics, fixed reduction orders, tickets reset by their last block: two independent learners fed the same
    batches end at bit-identical parameters and statistics after several graph replays; a longer run stays finite"""
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    finals, stats = [], []
    for _ in range(2):
        learner = H.make_learner(case, params, zstate)
        dbatch = learner._preprocess_batch_ppo(copy.deepcopy(batch))
        for it in range(4):
            st = dict(learner.learn(dbatch))
        finals.append({k: v.copy() for k, v in learner.model.numpy_params().items()})
        stats.append(st)
    for k in finals[0]:
        assert np.array_equal(finals[0][k], finals[1][k]), 'parameters differ run to run: ' + k
    for k in stats[0]:
        a, b = stats[0][k], stats[1][k]
        assert a == b or (np.isnan(a) and np.isnan(b)), 'statistic %s differs run to run: %r %r' % (k, a, b)
    for it in range(40):                                    # the last learner keeps going: counters, tickets, graphs
        st = learner.learn(dbatch)
    assert all(np.isfinite(v) for k, v in dict(st).items() if k != '_lr')


def test_gae_and_return_accessor():
    g, case = H.load_golden('ragged_clip')
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate)
    b = learner._preprocess_batch_ppo(copy.deepcopy(batch))
    adv, ret = learner._gae_and_return(b['obs'], b['obs_next'], b['rewards'], b['dones'])
    assert tuple(adv.shape) == g['advantages'].shape == (case['shape']['B'], 1)
    np.testing.assert_allclose(adv.cpu().numpy(), g['advantages'], atol=H.ATOL, rtol=H.RTOL)
    np.testing.assert_allclose(ret.cpu().numpy(), g['returns'], atol=H.ATOL, rtol=H.RTOL)


def test_rnn_three_learns_match_oracle():
    """consecutive learn() calls with the LSTM stem: both Adam states of the shared stem carry over"""
    import ppo_oracle
    g, case = H.load_golden('tiny_rnn_clip')
    batch, params, zstate = H.case_inputs(case)
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
    learner = H.make_learner(case, params, zstate)
    dbatch = learner._preprocess_batch_ppo(copy.deepcopy(batch))
    for it in range(3):
        so = O.learn(copy.deepcopy(batch))
        sl = learner.learn(dbatch)
        for k in so:
            if k != '_lr':
                at, rt = H.tol_for(k, H.ATOL, 2e-5)
                np.testing.assert_allclose(sl[k], so[k], atol=at, rtol=rt,
                                           err_msg='iteration %d stat %s' % (it, k))
    assert len(learner._graphs) == 1


def test_learns_on_moving_batches_use_staging_graph():
    """batches that arrive at a new address every time (what a sampling replay hands over): the
    learner captures once on its own staging buffers and replays -- same numbers as the oracle"""
    import ppo_oracle
    g, case = H.load_golden('cfg2_adapt')
    batch, params, zstate = H.case_inputs(case)
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
    learner = H.make_learner(case, params, zstate)
    captures, kept = [], []
    for it in range(4):
        so = O.learn(copy.deepcopy(batch))
        db = learner._preprocess_batch_ppo(copy.deepcopy(batch))
        kept.append(db)                                    # keep them alive: every batch at a new address
        sl = learner.learn(db)
        for k in so:
            if k != '_lr':
                at, rt = H.tol_for(k, H.ATOL, 2e-5)
                np.testing.assert_allclose(sl[k], so[k], atol=at, rtol=rt,
                                           err_msg='iteration %d stat %s' % (it, k))
        captures.append(id(next(iter(learner._graphs.values()))))
    assert learner._ws.staged is not None and len(learner._graphs) == 1
    assert captures[1] == captures[2] == captures[3]       # one staging-graph capture, then replays


def test_deferred_statistics_equal_the_synchronous_read():
    """learn() returns a mapping whose device -> host read-back resolves lazily (at the latest when
    the next learn() has been enqueued): every step's statistics, the trace and the counters must be
    what the synchronous read gives, in whatever order they are looked at"""
    from surreal_amd.learner.base import DeferredStats
    g, case = H.load_golden('cfg2_adapt')
    batch, params, zstate = H.case_inputs(case)
    runs = {}
    for lazy in (False, True):
        learner = H.make_learner(case, params, zstate, session_overrides={'lazy_stats': lazy})
        assert learner.lazy_stats == lazy
        out = [learner.learn(copy.deepcopy(batch)) for _ in range(3)]       # nothing looked at yet
        assert isinstance(out[0], DeferredStats) == lazy
        import pickle
        assert pickle.loads(pickle.dumps(out[0])) == dict(out[0])                 # travels as a plain dict
        last_trace, done = learner.trace, learner.epochs_executed             # resolves the last one
        runs[lazy] = ([dict(o) for o in out], last_trace, done, list(learner.kl_record),
                      dict(learner.tensorplex.latest), learner.model.actor_flat.cpu().clone())
    a, b = runs[False], runs[True]
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]
    assert torch.equal(a[5], b[5])
    assert a[0][0] != a[0][1]                                                 # three different steps


def test_fp64_arbiter_over_seeds_configs3():
    """BASELINE configs[3] at full size on six seeds (the golden's and five more, tests/golden/fp64_arbiter_seeds.json,
    each recorded from the REFERENCE'S OWN fp32 learner and from the float64 restatement).  On the extra seeds every
    loss / KL may be no further from float64 than SEED_LOSS_FACTOR x the reference's own fp32 value is (+ 1e-5; the
    reference does not reproduce itself to 1e-5 across two x86 hosts on this case -- tests/helpers.py -- while the
    golden seed's direct 1e-5 comparison stays in test_learner_matches_reference_golden_graph); the gradient norms'
    distances from float64 are held against the reference's over the same seeds as a distribution
    (helpers.assert_fp64_seed_distribution)."""
    doc = json.load(open(H.SEEDS_PATH))
    g0, case0 = H.load_golden(doc['base'])
    runs = [(case0, {'policy': json.loads(str(g0['policy_trace_json'])), 'value': json.loads(str(g0['value_trace_json']))},
             H.FP64['golden'][doc['base']], True)]
    for seed in sorted(doc['seeds'], key=int):
        rec = doc['seeds'][seed]
        runs.append((rec['case'], rec['reference_fp32'], rec['fp64'], False))
    hip = {'grad_norm_actor': [], 'grad_norm_critic': []}
    ref = {'grad_norm_actor': [], 'grad_norm_critic': []}
    worst = {}
    for case, ref32, f64, golden_seed in runs:
        batch, params, zstate = H.case_inputs(case)
        learner = H.make_learner(case, params, zstate)
        learner.learn(copy.deepcopy(batch))
        tr = learner.trace
        assert len(tr['policy']) == len(ref32['policy']) and len(tr['value']) == len(ref32['value'])
        for which in ('policy', 'value'):
            for k in ('_surr_loss', '_kl_loss_adapt', '_val_loss', '_pol_kl'):
                if k not in f64[which][0] or golden_seed:
                    continue
                # absolute distances (a surrogate loss sits near zero), the common 1e-5 abs + 1e-5 rel as the floor
                theirs = max(abs(a[k] - b[k]) for a, b in zip(ref32[which], f64[which]))
                for e, (a, b) in enumerate(zip(tr[which], f64[which])):
                    ours, bound = abs(a[k] - b[k]), H.SEED_LOSS_FACTOR * theirs + H.ATOL + H.RTOL * abs(b[k])
                    w = worst.get(k, (0, 0, 0, ''))
                    if ours / bound > w[0]:
                        worst[k] = (ours / bound, ours, theirs, '%s epoch %d' % (case['name'], e))
                    assert ours <= bound, '%s %s epoch %d: %.3g from float64, the reference at most %.3g (bound %.3g)' % (
                        case['name'], k, e, ours, theirs, bound)
        for which, key in (('policy', 'grad_norm_actor'), ('value', 'grad_norm_critic')):
            hip[key].append(H.seed_distances(tr[which], f64[which], key))
            ref[key].append(H.seed_distances(ref32[which], f64[which], key))
        del learner
        torch.cuda.empty_cache()
    H.assert_fp64_seed_distribution(hip, ref)
    for k, (share, ours, theirs, name) in worst.items():
        H.FP64_SEED_REPORT['loss ' + k] = {'worst_share_of_bound': share, 'hip_vs_fp64': ours, 'reference_vs_fp64': theirs,
                                            'case': name}


# ---- the fused forward + backward epoch on a device it does not have to itself ---------
def _learn_under_tenant(name, blocks, microseconds, session_overrides=None):
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate, session_overrides=session_overrides)
    db = learner._preprocess_batch_ppo(copy.deepcopy(batch))
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        learner.K.device_occupy(blocks, microseconds)
    stats = learner.learn(db)
    return g, case, learner, stats, db


def test_fused_fwdbwd_rides_out_a_co_resident_kernel():
    """cfg5 (1024 rows: 64 actor + 64 critic workgroups per epoch launch, every actor workgroup waiting inside the launch
    for the KL sums of all the others) while a tenant on another stream holds 192 of the 256 CUs for 4 ms -- longer than the
    whole learn.  The wait is a delay, not a failure: no sync error, results equal to the reference golden."""
    g, case, learner, stats, _ = _learn_under_tenant('cfg5_adapt', 192, 4000)
    assert learner._ws.fb, 'the fused forward + backward launch is the configuration under test'
    torch.cuda.synchronize()
    assert int(learner._ws.ctrl_i[L.C_SYNC_ERR].item()) == 0
    check_case('cfg5_adapt', g, case, learner, stats)


def test_shared_device_option_selects_the_two_launch_epochs():
    """session_config.learner.exclusive_device = False: no in-launch wait anywhere (two launches per epoch), same goldens"""
    g, case, learner, stats, _ = _learn_under_tenant('cfg5_adapt', 192, 4000, {'exclusive_device': False})
    assert not learner._ws.fb
    check_case('cfg5_adapt', g, case, learner, stats)


def test_fused_fwdbwd_timeout_fails_loudly_and_falls_back():
    """a tenant that keeps most CUs for longer than the wait's bound (0.25 s): the learn raises, the learner switches
    itself to the two-launch epochs, and -- parameters reloaded -- the same batch then gives the golden results"""
    name = 'cfg5_adapt'
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate)
    db = learner._preprocess_batch_ppo(copy.deepcopy(batch))
    learner.learn(db)                                 # the first learn() captures the graph (and would re-run a failed pass)
    torch.cuda.synchronize()
    for m in (learner.model, learner.ref_target_model):
        m.load_params(params)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        # 4 CUs left per XCD: 16 of the epoch launch's 64 actor workgroups are resident and wait for the other 48
        learner.K.device_occupy(224, 450000)
    with pytest.raises(RuntimeError, match='exclusive_device'):
        stats = learner.learn(db)
        dict(stats)                                   # (deferred statistics resolve here)
    torch.cuda.synchronize()
    assert learner._fb_timed_out
    fresh = H.make_learner(case, params, zstate)
    fresh._fb_timed_out = True
    stats = fresh.learn(copy.deepcopy(batch))
    assert not fresh._ws.fb
    check_case(name, g, case, fresh, stats)
