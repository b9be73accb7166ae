"""micro-benchmark of the persistent rollout kernel (GPU box): time per rollout at 1024 / 4096 actors and, with a
timing build (-DSMX_ROLLOUT_TIMING, loaded through SMX_LIB_PATH), where one step of a workgroup goes and the shader
clock the chip actually ran at (cycle counter against the 100 MHz wall clock).
    python scripts/bench_rollout.py [actors ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from surreal_amd import _lib as L
from surreal_amd.kernels import HipKernels
from surreal_amd.agent import PPOAgent
from surreal_amd.env import SyntheticVecEnv
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config

K = HipKernels()
D, A, T = 376, 17, 128
lc = ppo_learner_config(); lc.algo.rnn.if_rnn_policy = False
agent = PPOAgent(lc, ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_bench_rollout'), agent_id=0, agent_mode='training')
timing = hasattr(K.lib, 'smx_rollout_debug_tbuf')
for n in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    venv = SyntheticVecEnv(n, D, A, episode_len=T)
    eps = torch.randn(T, n, A, device='cuda')
    tb = torch.zeros(((n + 3) // 4) * 16, dtype=torch.int64, device="cuda")     # (one row of 16 stamps per workgroup; 4 actors each at most)
    if timing:
        K.lib.smx_rollout_debug_tbuf.argtypes = [ctypes.c_void_p]
        K.lib.smx_rollout_debug_tbuf(ctypes.c_void_p(tb.data_ptr()))

    def once():
        venv.reset(); venv.start_rollout(T, info_width=2 * A); venv.rollout(agent, eps=eps)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        venv.reset(); venv.slot = 0; venv.rollout(agent, eps=eps)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('%5d actors x %d steps: %.3f ms per rollout = %.1f us per step, %.3g env-steps/s' % (n, T, ms, ms / T * 1e3, n * T / ms * 1e3))
    if timing:
        t = tb.view(-1, 16).cpu().double()
        ghz = ((t[:, 15] - t[:, 13]) / ((t[:, 14] - t[:, 12]) * 10.0)).mean()       # cycles per ns
        d = t[:, 1:6] - t[:, 0:5]
        print('   shader clock %.2f GHz;  one step of a workgroup (cycles, mean over workgroups): layer1 %.0f  layer2 %.0f  '
              'layer3 %.0f  head %.0f  env step + record + z-filter %.0f  = %.0f' % ((ghz,) + tuple(d.mean(0).tolist()) + (float((t[:, 5] - t[:, 0]).mean()),)))
