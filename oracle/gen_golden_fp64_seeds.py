#!/usr/bin/env python
"""
TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).  tests/golden/fp64_arbiter_seeds.json: BASELINE
configs[3] at its stated size (256 actors x 32 steps of 3 x 84 x 84 uint8 frames + 32-d state, CNN + LSTM policy --
the case of tests/golden/ppo_cfg4_pixel_rnn_256x32.npz) on FIVE MORE seeds of inputs and parameters, each run twice:
  * by the REFERENCE'S OWN learner in fp32 (surreal/learner/ppo.py under oracle/ref_shims.py, as oracle/gen_golden.py
    runs it) -> its gradient-norm traces and losses;
  * by the restatement (oracle/ppo_oracle.py) in FLOAT64 -> the same quantities to ~1e-12.
What it is for: on this case a gradient norm is a sum of 7168 nearly cancelling row terms behind a randomly initialised
stem, and ONE ReLU pre-activation within fp32 rounding of zero decides a percent of it.  How far a
legitimate fp32 evaluation lands from exact arithmetic is therefore a DISTRIBUTION over inputs, not a number: one seed
(round 3) gave the reference 2.0e-2 and the HIP path 5.8e-2, and a bound fitted to that one sample passed at 98.8 %.
The tests now hold the HIP path's distances over the seeds against the reference's over the same seeds
(tests/helpers.py::assert_fp64_seed_distribution).

    python oracle/gen_golden_fp64_seeds.py [seed ...]        (~3 minutes of CPU per seed on 8 threads)
"""
import copy
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import ref_shims  # noqa: E402
import gen_golden as G  # noqa: E402
import ppo_oracle  # noqa: E402

BASE = 'cfg4_pixel_rnn_256x32'
SEEDS = [11, 12, 13, 14, 15]
KEYS = ('grad_norm_actor', 'grad_norm_critic', '_surr_loss', '_kl_loss_adapt', '_val_loss', '_pol_kl')
PATH = os.path.join(ROOT, 'tests', 'golden', 'fp64_arbiter_seeds.json')


def slim(rows):
    return [{k: float(r[k]) for k in KEYS if k in r} for r in rows]


def case_for(seed):
    case = copy.deepcopy(G.CASES[BASE])
    case['name'] = '%s_seed%d' % (BASE, seed)
    case['batch_args'] = dict(seed=100 + seed)
    case['param_args'] = dict(seed=200 + seed)
    case['z_args'] = dict(seed=300 + seed)
    case['hyper'] = dict(case['hyper'], n_step=case['shape']['N'])
    return case


def main(seeds):
    ref = ref_shims.import_reference()
    doc = json.load(open(PATH)) if os.path.exists(PATH) else {'base': BASE, 'seeds': {}}
    doc['note'] = ('per seed: the reference\'s own fp32 learner and the float64 restatement on the same inputs '
                   '(oracle/gen_golden_fp64_seeds.py); case = tests/golden/ppo_%s.npz with other generator seeds' % BASE)
    for seed in seeds:
        case = case_for(seed)
        torch.set_default_dtype(torch.float32)
        t0 = time.time()
        batch, params, zstate, trace, stats, final, zfinal, L = G.run_reference(ref, case)
        t_ref = time.time() - t0
        torch.set_default_dtype(torch.float64)
        t0 = time.time()
        shp = case['shape']
        O = ppo_oracle.OraclePPOLearner(params, shp['A'], shp['B'], zstate=zstate, **case['hyper'])
        O.learn(copy.deepcopy(batch))
        t_64 = time.time() - t0
        torch.set_default_dtype(torch.float32)
        doc['seeds'][str(seed)] = {
            'case': case,
            'reference_fp32': {'policy': slim(trace['policy']), 'value': slim(trace['value'])},
            'fp64': {'policy': slim(O.trace['policy']), 'value': slim(O.trace['value'])},
            'seconds': {'reference_fp32': t_ref, 'fp64': t_64}}
        d = {}
        for which, key in (('policy', 'grad_norm_actor'), ('value', 'grad_norm_critic')):
            a, b = doc['seeds'][str(seed)]['reference_fp32'][which], doc['seeds'][str(seed)]['fp64'][which]
            d[key] = max(abs(x[key] - y[key]) / abs(y[key]) for x, y in zip(a, b))
        print('seed %d: reference %.0f s, float64 %.0f s; reference vs float64: %s' % (seed, t_ref, t_64, d), flush=True)
        json.dump(doc, open(PATH, 'w'), indent=0)
    print('wrote', PATH, os.path.getsize(PATH), 'bytes')


if __name__ == '__main__':
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    main([int(a) for a in sys.argv[1:]] or SEEDS)
