"""How reproducible is the REFERENCE ITSELF across hosts over a multi-learn sequence?  Runs the
oracle restatement (bit-identical to the reference in the build container, asserted by
oracle/gen_golden_sequence.py) on THIS host's CPU and prints its gradient norms next to the golden
values recorded in the build container, and next to the HIP path's."""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import ppo_oracle  # noqa: E402
import sequence_cases as SC  # noqa: E402
from surreal_amd import synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5_publish_adapt'
case, records = SC.DOC[name]['case'], SC.DOC[name]['records']
shp = case['shape']
hyper = dict(case['hyper'])
hyper['n_step'] = shp['N']
params = synthetic.make_ppo_params(shp['D'], shp['A'], hidden=tuple(case['hidden']), rnn_hidden=case['rnn_hidden'],
                                   **case['param_args'])
zstate = synthetic.make_zfilter_state(shp['D'], **case['z_args'])
O = ppo_oracle.OraclePPOLearner(params, shp['A'], shp['B'], zstate=zstate, **hyper)
L = SC.make_learner(case) if torch.cuda.is_available() else None
for r in records:
    if r['op'] == 'learn':
        b = synthetic.make_ppo_batch(shp['B'], shp['N'], shp['D'], shp['A'], rnn_hidden=case['rnn_hidden'],
                                     seed=r['seed'], **case['batch_args'])
        st = O.learn(copy.deepcopy(b))
        hs = dict(L.learn(copy.deepcopy(b))) if L else {}
        for k in ('grad_norm_critic', 'grad_norm_actor', '_val_loss', '_kl_loss_adapt', '_pol_kl'):
            if k in st:
                print('%-18s golden %.7f  oracle(this host) %.7f (rel %.1e)  hip %s' % (
                    k, r['stats'][k], st[k], abs(st[k] - r['stats'][k]) / abs(r['stats'][k]),
                    ('%.7f (rel %.1e)' % (hs[k], abs(hs[k] - r['stats'][k]) / abs(r['stats'][k]))) if hs else '-'))
        print()
    else:
        if O.exp_counter >= case['exp_interval']:
            O._post_publish()
        if L:
            L.publish_parameter(0)
