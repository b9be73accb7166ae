"""print the per-epoch trace of one golden case on the HIP path next to the reference's golden (GPU box);
`--oracle`: also the oracle restatement (= the reference's ATen ops) on THIS host's CPU, i.e. how far the reference
itself drifts between the host the golden was recorded on and this one.
    python tests/diag/diag_case.py cfg4_pixel_rnn_256x32 [--oracle]"""
import sys, os, json, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers as H
name = sys.argv[1]
g, case = H.load_golden(name)
batch, params, zstate = H.case_inputs(case)
learner = H.make_learner(case, params, zstate)
stats = learner.learn(copy.deepcopy(batch))
for which, key in (('policy', 'policy_trace_json'), ('value', 'value_trace_json')):
    ref = json.loads(str(g[key]))
    for e, (a, b) in enumerate(zip(learner.trace[which], ref)):
        print(which, e, '  '.join('%s %.6g/%.6g (%.1e)' % (k, a[k], b[k], abs(a[k] - b[k]) / (abs(b[k]) + 1e-12)) for k in b))
if '--oracle' in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ppo_oracle
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
    O.learn(batch)
    for which, key in (('policy', 'policy_trace_json'), ('value', 'value_trace_json')):
        ref = json.loads(str(g[key]))
        for e, (a, b, c) in enumerate(zip(O.trace[which], ref, learner.trace[which])):
            print('oracle-here', which, e, '  '.join('%s here %.6g golden %.6g hip %.6g' % (k, a[k], b[k], c[k])
                                                    for k in b if 'grad_norm' in k or 'loss' in k))
    ck = json.loads(str(g['final_checksum_json']))
    oh, hp = O.model.numpy_params(), learner.model.numpy_params()
    for k, (s_, sq) in ck.items():
        so = float(np.sum(oh[k].astype(np.float64) ** 2)); sh = float(np.sum(hp[k].astype(np.float64) ** 2))
        print('checksum %-22s golden %.6g  oracle-here rel %.1e  hip rel %.1e' % (k, sq, abs(so - sq) / sq, abs(sh - sq) / sq))
