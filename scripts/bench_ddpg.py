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
# how much of that is the HOST: the same loop's enqueue time alone (the clock stops before the device is waited for), and the
# device's own time per iteration from events around the loop
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for i in range(n): st = L.learn(batches[i % 8])
e1.record(); th = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
print('   host enqueue time per learn(): %.3f ms; device time per iteration (events): %.3f ms' % (th * 1e3, e0.elapsed_time(e1) / n))
# ---- with the uniform replay in the loop: 1e6 SSAR rows resident in HBM, sample 512 + learn ----
from surreal_amd.replay import UniformReplay
lc.replay.memory_size = 1000000
R = UniformReplay(lc, ddpg_env_config(D, A), ddpg_session_config())
g = torch.Generator(device='cuda').manual_seed(0)
for _ in range(10):
    n = 100000
    R.insert_batch({'obs': torch.randn(n, D, device='cuda', generator=g), 'obs_next': torch.randn(n, D, device='cuda', generator=g),
                    'actions': torch.rand(n, A, device='cuda', generator=g) * 2 - 1, 'rewards': torch.randn(n, device='cuda', generator=g),
                    'dones': (torch.rand(n, device='cuda', generator=g) < 0.01).float()})


def sample_and_learn():
    f = R.sample_batch(B)
    return L.learn({'obs': {'low_dim': {'flat_inputs': f['obs']}}, 'obs_next': {'low_dim': {'flat_inputs': f['obs_next']}},
                    'actions': f['actions'], 'rewards': f['rewards'].view(B, 1), 'dones': f['dones'].view(B, 1)})


for i in range(20): sample_and_learn()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 300
for i in range(n): st = sample_and_learn()
torch.cuda.synchronize(); ds = (time.perf_counter() - t0) / n
print('DDPG sample(512 of 1e6) + learn: %.3f ms/iter  %.3g samples/s  (hipGraph %s)' % (ds * 1e3, B / ds, L._ws.graph is not None))
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
