#!/usr/bin/env python
"""
TEST INFRASTRUCTURE ONLY.  tests/golden/fp64_arbiter.json: the PPO restatement (oracle/ppo_oracle.py,
asserted bit-identical to the reference's own learner in fp32 by oracle/gen_golden.py) run in FLOAT64
on the inputs of the goldens whose gradient norms are NOT reproducible to 1e-5 in fp32 -- by the
reference itself, between two x86 hosts (tests/helpers.py).

What it arbitrates: a gradient norm is a sum of thousands of nearly cancelling row terms; ATen's fp32
summation order and the HIP kernels' (split-K, implicit-GEMM convolutions) are both legitimate fp32
evaluations of the same real number.  The float64 run is that number to ~1e-12.  The tests then hold the
HIP path to  |HIP - fp64| <= 2 |ATen-fp32 - fp64| + floor  over each trace (tests/helpers.py::
assert_fp64_arbiter): the HIP gradients may not be further from exact arithmetic than the reference's
own fp32 path is -- a bound that does not move with the host the golden was recorded on.

Same seeded inputs and injected parameters as the goldens (they are float32 numpy arrays, widened
exactly).  Runs in the build container or anywhere else: it does not need /root/reference.
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H  # noqa: E402
import ppo_oracle  # noqa: E402
from surreal_amd import synthetic  # noqa: E402

GOLDEN_CASES = ['cfg4_pixel_rnn_256x32', 'cfg4_pixel_adapt', 'cfg5_clip', 'cfg5_adapt', 'cfg2_adapt', 'cfg1_rnn_adapt',
                # round 6: the LSTM policy at the shapes bench.py prices (gradient norms of ~5e-3 summed over 126 976 rows)
                'cfg2_rnn_adapt', 'cfg2_rnn_clip', 'cfg5_rnn_adapt', 'cfg5_rnn_clip', 'b1024_d17_rnn_adapt']
SEQUENCE_CASES = ['cfg5_publish_adapt', 'publish_rnn_adapt']


def rows(trace):
    return {k: [{kk: float(vv) for kk, vv in r.items()} for r in trace[k]] for k in ('policy', 'value')}


def run_golden(name):
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    shp = case['shape']
    hyper = dict(case['hyper'])
    hyper['n_step'] = shp['N']
    O = ppo_oracle.OraclePPOLearner(params, shp['A'], shp['B'], zstate=zstate, **hyper)
    stats = O.learn(copy.deepcopy(batch))
    out = rows(O.trace)
    out['stats'] = {k: float(v) for k, v in stats.items()}
    return out


def run_sequence(name):
    import sequence_cases as SC
    case, records = SC.DOC[name]['case'], SC.DOC[name]['records']
    shp = case['shape']
    O = SC.make_oracle(case)
    learns = []
    for r in records:
        if r['op'] == 'learn':
            batch = synthetic.make_ppo_batch(shp['B'], shp['N'], shp['D'], shp['A'], rnn_hidden=case['rnn_hidden'],
                                             seed=r['seed'], **case['batch_args'])
            stats = O.learn(copy.deepcopy(batch))
            d = rows(O.trace)
            d['stats'] = {k: float(v) for k, v in stats.items()}
            learns.append(d)
        elif O.exp_counter >= case['exp_interval']:
            O._post_publish()
    return learns


def main(only=None):
    """`only`: case names to (re)compute; the others are kept from the committed file"""
    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    path = os.path.join(ROOT, 'tests', 'golden', 'fp64_arbiter.json')
    out = {'golden': {}, 'sequences': {},
           'note': 'float64 run of oracle/ppo_oracle.py on the goldens\' inputs (oracle/gen_golden_fp64.py)'}
    if only:
        out = json.load(open(path))
    for name in GOLDEN_CASES:
        if only and name not in only:
            continue
        t0 = time.time()
        out['golden'][name] = run_golden(name)
        v = out['golden'][name]['value']
        print('%-26s %5.1f s   grad_norm_critic[0] = %.9g' % (name, time.time() - t0, v[0].get('grad_norm_critic', 0)))
    for name in SEQUENCE_CASES:
        if only and name not in only:
            continue
        t0 = time.time()
        out['sequences'][name] = run_sequence(name)
        print('%-26s %5.1f s   %d learns' % (name, time.time() - t0, len(out['sequences'][name])))
    with open(path, 'w') as fp:
        json.dump(out, fp, indent=0)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main(sys.argv[1:] or None)
