#!/usr/bin/env python
"""
TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).  Times the REFERENCE'S OWN learner
(surreal/learner/ppo.py: _preprocess_batch_ppo + _optimize, run under oracle/ref_shims.py exactly as
oracle/gen_golden.py runs it) beside the CPU restatement bench.py uses as `cpu_baseline` (oracle/ppo_oracle.py,
kind "port") on the SAME host, batch, parameters and thread count, for every BASELINE configuration the bench
prices.  bench.py cannot run the reference itself -- /root/reference does not exist on the GPU box -- so this record
is what says whether the port flatters the GPU (it does not: the port is the faster of the two).

    python oracle/time_reference_vs_port.py            -> profiles/r04_cpu_reference_vs_port.json
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

import ref_shims  # noqa: E402
from surreal_amd import synthetic  # noqa: E402
import ppo_oracle  # noqa: E402
import gen_golden as G  # noqa: E402

CONFIGS = [
    # name, shape, hyper, rnn_hidden, timed learns
    ('configs[4] PPO 1024x128x376 MLP, adapt', dict(B=1024, N=128, D=376, A=17), dict(ppo_mode='adapt', kl_target=1e9), 0, 3),
    ('configs[1] PPO 64x128 D=17 A=6 MLP, adapt', dict(B=64, N=128, D=17, A=6), dict(ppo_mode='adapt', kl_target=1e9), 0, 5),
    ('configs[1] PPO 64x128 D=17 A=6 LSTM(100, H=5), adapt', dict(B=64, N=128, D=17, A=6),
     dict(ppo_mode='adapt', kl_target=1e9, if_rnn_policy=True, horizon=5), 100, 3),
    ('configs[0] PPO 2x25 D=17 A=6 LSTM(100, H=5), adapt (test_ppo_gym --unit-test shape)', dict(B=2, N=25, D=17, A=6),
     dict(ppo_mode='adapt', kl_target=1e9, if_rnn_policy=True, horizon=5), 100, 10),
]


def timed(fn, n):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
    return ts


def main():
    ref = ref_shims.import_reference()
    threads = int(os.environ.get('SMX_THREADS', str(os.cpu_count() or 1)))
    torch.set_num_threads(threads)
    out = {'host': os.uname().nodename, 'threads': threads, 'torch': torch.__version__, 'rows': []}
    try:
        import subprocess
        for ln in subprocess.run(['lscpu'], capture_output=True, text=True).stdout.splitlines():
            if ln.startswith('Model name'):
                out['cpu_model'] = ln.split(':', 1)[1].strip()
    except Exception:
        pass
    for name, shp, hyper, rnn_hidden, n in CONFIGS:
        B, N, D, A = shp['B'], shp['N'], shp['D'], shp['A']
        batch = synthetic.make_ppo_batch(B, N, D, A, seed=100, rnn_hidden=rnn_hidden)
        params = synthetic.make_ppo_params(D, A, hidden=(300, 200), seed=1, rnn_hidden=rnn_hidden)
        zstate = synthetic.make_zfilter_state(D, seed=2)
        h = dict(hyper, n_step=N)
        Lr = G.build_reference_learner(ref, params, zstate, B, N, D, A, h)

        def ref_learn():
            bd = ref_shims.BeneDict(copy.deepcopy(batch))
            bd = Lr._preprocess_batch_ppo(bd)
            Lr._optimize(bd.obs, bd.actions, bd.rewards, bd.obs_next, bd.persistent_infos, bd.onetime_infos, bd.dones)
        O = ppo_oracle.OraclePPOLearner(params, A, B, zstate=zstate, **h)
        tr = timed(ref_learn, n)
        tp = timed(lambda: O.learn(copy.deepcopy(batch)), n)
        row = {'config': name, 'B': B, 'N': N, 'learns_timed': n,
               'reference_s_per_learn': sum(tr) / n, 'reference_min_s': min(tr),
               'port_s_per_learn': sum(tp) / n, 'port_min_s': min(tp),
               'port_over_reference_time': (sum(tp) / n) / (sum(tr) / n),
               'reference_env_steps_per_s': B * N / (sum(tr) / n), 'port_env_steps_per_s': B * N / (sum(tp) / n)}
        print(json.dumps(row), flush=True)
        out['rows'].append(row)
    out['what'] = ('reference = surreal/learner/ppo.py PPOLearner._preprocess_batch_ppo + _optimize under the shims of '
                   'oracle/ref_shims.py; port = oracle/ppo_oracle.py OraclePPOLearner.learn (bench.py cpu_baseline, kind '
                   '"port"); one warm-up call, then `learns_timed` calls each, same process, KL early exit disabled')
    path = os.path.join(ROOT, 'profiles', 'r04_cpu_reference_vs_port.json')
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
