"""one epoch of a golden case: the HIP learner's critic / actor gradients (the buffers clip-norm + Adam read) against the
oracle's autograd gradients on THIS host, element by element -- where they differ, which rows / columns of the layer.
    python tests/diag/diag_epoch0_grads.py cfg5_clip [key=value session options ...]     (GPU box)"""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import helpers as H
import ppo_oracle

name = sys.argv[1]
opts = {}
for a in sys.argv[2:]:
    k, v = a.split('=')
    opts[k] = {'True': True, 'False': False}.get(v, v)
g, case = H.load_golden(name)
case = copy.deepcopy(case)
case['hyper'].update(epoch_policy=1, epoch_baseline=1)
batch, params, zstate = H.case_inputs(case)
learner = H.make_learner(case, params, zstate, session_overrides=opts)
learner.learn(copy.deepcopy(batch))
hyper = dict(case['hyper']); hyper['n_step'] = case['shape']['N']
O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
O.learn(copy.deepcopy(batch))
ws, m = learner._ws, learner.model
named = m.named_parameters()
print('--- %s %s: gradients of the one epoch' % (name, opts))
for flat, gbuf, pre in ((m.actor_flat, ws.grads_a, 'actor.'), (m.critic_flat, ws.grads_c, 'critic.')):
    for k, v in named.items():
        if not k.startswith(pre) or O.model.p[k].grad is None:
            continue
        off = (v.data_ptr() - flat.data_ptr()) // 4
        if not (0 <= off and off + v.numel() <= gbuf.numel()):
            continue
        gh = gbuf[off:off + v.numel()].view(v.shape).cpu().numpy()
        go = O.model.p[k].grad.detach().numpy()
        d = np.abs(gh - go)
        scale = np.abs(go).max()
        bad = d > 1e-6 * scale + 1e-9
        line = '%-16s |g|max %.3g  max diff %.3g (%.2g of |g|max)  elements off by > 1e-6 |g|max: %d of %d' % (
            k, scale, d.max(), d.max() / scale, bad.sum(), d.size)
        if bad.any() and d.ndim == 2:
            r, c = np.nonzero(bad)
            line += '\n      rows (output units) %s\n      cols (inputs) %d distinct; diff on the worst row: median %.3g max %.3g' % (
                sorted(set(r.tolist()))[:20], len(set(c.tolist())), np.median(d[r[0]]), d[r[0]].max())
        print(line)
