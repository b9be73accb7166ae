"""time PPOLearner.learn() on BASELINE configs[3] shapes (cfg 4: pixel observations, 256 actors, CNN
stem 16c8s4-32c4s2-FC256 on 3x84x84 uint8 frames + 32-d robot state, A = 8)"""
import sys, os, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from surreal_amd import synthetic
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
from surreal_amd.learner.ppo import PPOLearner


def run(B, N, D=32, A=8, rnn=True, steps=3, pixel=(3, 84, 84)):
    lc = ppo_learner_config()
    lc.algo.n_step = N; lc.algo.stride = N
    lc.algo.rnn.if_rnn_policy = rnn
    lc.algo.consts.kl_target = 1e9
    lc.replay.batch_size = B
    L = PPOLearner(lc, ppo_env_config(D, A, pixel=pixel), ppo_session_config('/tmp/x'))
    batch = synthetic.make_ppo_batch(B, N, D, A, seed=1, rnn_hidden=lc.algo.rnn.rnn_hidden if rnn else 0,
                                     pixel=pixel)
    db = L._preprocess_batch_ppo(copy.deepcopy(batch))
    for _ in range(2): L.learn(db)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): st = L.learn(db)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    frames = B * (N + 1)
    print('pixel B=%4d N=%3d rnn=%s: %.1f ms/learn  %.3g env-steps/s  (%d frames resident = %.0f MB uint8; surr %.4f)'
          % (B, N, rnn, dt * 1e3, B * N / dt, frames, frames * 3 * 84 * 84 / 1e6, st['_surr_loss']))


if __name__ == '__main__':
    run(256, 32, rnn=True)
    run(256, 32, rnn=False)
    if len(sys.argv) > 1:
        run(256, 128, rnn=True, steps=2)
