"""DDPG secondary metric: samples/sec = 512 / wall(sample + learn) at
BASELINE configs[2] (HalfCheetah shapes D=17, A=6, uniform replay, batch 512), device-resident
replay shard; and the oracle (reference ATen path) on the host CPU beside it."""
import sys, os, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch
from surreal_amd import synthetic
from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config
from surreal_amd.learner.ddpg import DDPGLearner

B, D, A = 512, 17, 6
lc = ddpg_learner_config(); lc.replay.batch_size = B
L = DDPGLearner(lc, ddpg_env_config(D, A), ddpg_session_config())
batches = [L.preprocess(synthetic.make_ddpg_batch(B, D, A, seed=s)) for s in range(8)]
for i in range(20): L.learn(batches[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 300
for i in range(n): st = L.learn(batches[i % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print('DDPG learn (batch resident): %.3f ms/iter  %.3g samples/s  critic_loss %.4f' % (dt * 1e3, B / dt, st['critic_loss']))
try:
    import ddpg_oracle
    params = ddpg_oracle.make_ddpg_params(D, A, (300, 200), (400, 300), seed=3)
    O = ddpg_oracle.OracleDDPGLearner(params, A) if hasattr(ddpg_oracle, 'OracleDDPGLearner') else None
    if O is not None:
        hb = [synthetic.make_ddpg_batch(B, D, A, seed=s) for s in range(8)]
        for i in range(5): O.learn(copy.deepcopy(hb[i % 8]))
        t0 = time.perf_counter(); n = 100
        for i in range(n): O.learn(copy.deepcopy(hb[i % 8]))
        dc = (time.perf_counter() - t0) / n
        print('oracle (reference ATen path, %d threads): %.3f ms/iter  %.3g samples/s  -> x%.1f' % (torch.get_num_threads(), dc * 1e3, B / dc, dc / dt))
except Exception as e:
    print('oracle timing skipped:', repr(e))
