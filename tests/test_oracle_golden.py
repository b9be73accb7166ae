"""CPU tier: the oracle restatement (oracle/ppo_oracle.py) must reproduce every golden vector
that oracle/gen_golden.py recorded from the REFERENCE's own code (bit-for-bit at generation
time; a small tolerance here absorbs BLAS differences between hosts)."""
import json

import numpy as np
import pytest

import helpers as H
import ppo_oracle


@pytest.mark.parametrize('name', H.golden_cases(big=False))
def test_oracle_reproduces_reference_golden(name):
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate,
                                    **hyper)
    stats = O.learn(batch)
    np.testing.assert_allclose(O.trace['advantages'], g['advantages'], atol=2e-6, rtol=1e-6)
    np.testing.assert_allclose(O.trace['returns'], g['returns'], atol=5e-6, rtol=1e-6)
    vr = O.trace['values_raw']
    np.testing.assert_allclose(vr[:g['values_raw'].shape[0]], g['values_raw'], atol=5e-6, rtol=1e-6)
    H.assert_trace_close(O.trace, g, atol=2e-6, rtol=2e-6, what=name)
    H.assert_stats_close(stats, g, atol=2e-6, rtol=2e-6, what=name)
    if zstate is not None:
        for k in ('running_sum', 'running_sumsq', 'count'):
            np.testing.assert_allclose(O.model.z_filter.state()[k], g['zfinal.' + k], rtol=1e-6)


def test_golden_covers_the_branches():
    """the fixture set must exercise: both ppo modes, the adapt KL-cutoff penalty, the KL early
    exit, dones inside the window, off-policy likelihood clamps, no-z-filter / no-norm."""
    names = H.golden_cases()
    assert any('clip' in n for n in names) and any('adapt' in n for n in names)
    g, _ = H.load_golden('cfg5_adapt_earlyexit')
    tr = json.loads(str(g['policy_trace_json']))
    assert 1 < len(tr) < 10                                  # early exit fired mid-way
    assert any(abs(t['_kl_loss_adapt'] - t['_surr_loss']) > 1e-4 for t in tr)
    g, _ = H.load_golden('tiny_adapt_cutoff')
    assert len(json.loads(str(g['policy_trace_json']))) == 1  # exit after the first update
