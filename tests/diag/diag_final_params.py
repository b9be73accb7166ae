"""where the updated parameters of a golden case leave the reference's: the HIP learner against the oracle restatement run
on THIS host, for 1, 2, 3 and all epochs (epoch_policy = epoch_baseline = e), per tensor -- and, after ONE Adam step
(update = lr * sign(g) unless |g| ~ eps), the oracle's gradient magnitude at the elements that moved the other way.
    python tests/diag/diag_final_params.py cfg5_clip [key=value session options ...]     (GPU box)"""
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
g, case0 = H.load_golden(name)
for epochs in (1, 2, 3, 10):
    case = copy.deepcopy(case0)
    case['hyper'].update(epoch_policy=epochs, epoch_baseline=epochs)
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate, session_overrides=opts)
    learner.learn(copy.deepcopy(batch))
    hyper = dict(case['hyper']); hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
    O.learn(copy.deepcopy(batch))
    hp, op = learner.model.numpy_params(), O.model.numpy_params()
    print('--- %s %s epochs = %d' % (name, opts, epochs))
    for k in op:
        d = np.abs(hp[k] - op[k])
        line = '%-18s max %.3g  frac > 1e-6: %.4f  > 1e-5: %.4f' % (k, d.max(), np.mean(d > 1e-6), np.mean(d > 1e-5))
        gr = O.model.p[k].grad
        if epochs == 1 and gr is not None and (d > 1e-6).any():
            ga = gr.detach().abs().numpy()
            line += '   |g| at the differing elements: median %.3g max %.3g  (all elements: median %.3g)' % (
                np.median(ga[d > 1e-6]), ga[d > 1e-6].max(), np.median(ga))
        print(line)
