"""diagnostic (GPU box): locate where the critic update of cfg5 departs from the oracle"""
import copy, json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as H
import ppo_oracle

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5_clip'
g, case = H.load_golden(name)
batch, params, zstate = H.case_inputs(case)
case['hyper']['epoch_baseline'] = int(sys.argv[2]) if len(sys.argv) > 2 else 1
case['hyper']['epoch_policy'] = 1
hyper = dict(case['hyper']); hyper['n_step'] = case['shape']['N']
O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
# capture oracle critic grads
O.learn(copy.deepcopy(batch))
og = {k: v.grad.detach().numpy().copy() for k, v in O.model.p.items() if v.grad is not None}
op = O.model.numpy_params()
L = H.make_learner(case, params, zstate, session_overrides={'use_hip_graph': False, 'overlap_value_epochs': False})
L.learn(copy.deepcopy(batch))
gp = L.model.numpy_params()
ws = L._ws
gc = ws.grads_c.cpu().numpy()
net = L.model.critic
o = 0
for nm, key in (('W1', 'critic.fc1.W'), ('b1', 'critic.fc1.b'), ('W2', 'critic.fc2.W'), ('b2', 'critic.fc2.b'), ('W3', 'critic.fc3.W'), ('b3', 'critic.fc3.b')):
    n = net.views[nm].numel()
    gg = gc[o:o + n].reshape(net.views[nm].shape); o += n
    d = np.abs(gg - og[key])
    pd = np.abs(gp[key] - op[key])
    p0 = params[key]
    print('%-14s grad: max|d|=%.3e max|g|=%.3e  n(|d|>1e-6)=%d   param: max|d|=%.3e n(>1e-6)=%d  upd_oracle max=%.3e' % (
        key, d.max(), np.abs(og[key]).max(), (d > 1e-6).sum(), pd.max(), (pd > 1e-6).sum(), np.abs(op[key]-p0).max()))
    if (pd > 1e-6).sum():
        idx = np.argwhere(pd > 1e-6)[:5]
        for ix in idx:
            ix = tuple(ix)
            print('     at', ix, 'grad gpu %.4e oracle %.4e | param gpu %.8f oracle %.8f init %.8f' % (gg[ix], og[key][ix], gp[key][ix], op[key][ix], p0[ix]))
print('value stats gpu', L.trace['value'], 'oracle', O.trace['value'])
