import copy, json, sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as H
import ppo_oracle
g, case = H.load_golden('cfg5_clip')
batch, params, zstate = H.case_inputs(case)
hyper = dict(case['hyper']); hyper['n_step'] = case['shape']['N']; hyper['epoch_baseline'] = 1; hyper['epoch_policy'] = 1
O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
O.learn(copy.deepcopy(batch))
p1 = O.model.numpy_params()
for k in p1:
    if not k.startswith('critic'): continue
    u = (p1[k].astype(np.float64) - params[k]) / 1e-4
    gr = O.model.p[k].grad.numpy()
    print('%-14s n=%6d  sum(sign)=%+d  n(|u|<0.99)=%d  sum(u)=%.6f  md5(sign)=%s  min|g|=%.2e n(|g|<1e-7)=%d sum(g)=%.9e' % (
        k, u.size, int(np.sign(u).sum()), int((np.abs(u) < 0.99).sum()), u.sum(),
        hashlib.md5(np.sign(u).astype(np.int8).tobytes()).hexdigest()[:8], np.abs(gr).min(), int((np.abs(gr) < 1e-7).sum()), gr.astype(np.float64).sum()))
