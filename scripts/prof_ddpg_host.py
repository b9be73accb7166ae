"""where the HOST time of DDPGLearner.learn() goes (cProfile over 3000 graph-replay iterations at configs[2] shapes)"""
import sys, os, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from surreal_amd import synthetic
from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config
from surreal_amd.learner.ddpg import DDPGLearner
from surreal_amd.replay import UniformReplay

B, D, A = 512, 17, 6
lc = ddpg_learner_config(); lc.replay.batch_size = B; lc.replay.memory_size = 100000
L = DDPGLearner(lc, ddpg_env_config(D, A), ddpg_session_config())
R = UniformReplay(lc, ddpg_env_config(D, A), ddpg_session_config())
g = torch.Generator(device='cuda').manual_seed(0)
n = 100000
R.insert_batch({'obs': torch.randn(n, D, device='cuda', generator=g), 'obs_next': torch.randn(n, D, device='cuda', generator=g),
                'actions': torch.rand(n, A, device='cuda', generator=g) * 2 - 1, 'rewards': torch.randn(n, device='cuda', generator=g),
                'dones': (torch.rand(n, device='cuda', generator=g) < 0.01).float()})


def step():
    f = R.sample_batch(B, out=L.staging_fields(B)) if os.environ.get('SMX_STAGED', '1') == '1' else R.sample_batch(B)
    return L.learn({'obs': {'low_dim': {'flat_inputs': f['obs']}}, 'obs_next': {'low_dim': {'flat_inputs': f['obs_next']}},
                    'actions': f['actions'], 'rewards': f['rewards'].view(B, 1), 'dones': f['dones'].view(B, 1)})


for _ in range(50): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000): step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
