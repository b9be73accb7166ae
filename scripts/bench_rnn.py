"""time PPOLearner.learn() with the LSTM stem (reference default policy) at a few shapes"""
import sys, os, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from surreal_amd import synthetic
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
from surreal_amd.learner.ppo import PPOLearner

def run(B, N, D, A, HID=100, horizon=5, steps=5):
    lc = ppo_learner_config()
    lc.algo.n_step = N; lc.algo.stride = N
    lc.algo.rnn.if_rnn_policy = True; lc.algo.rnn.rnn_hidden = HID; lc.algo.rnn.horizon = horizon
    lc.algo.consts.kl_target = 1e9
    lc.replay.batch_size = B
    sc = ppo_session_config('/tmp/x')
    import os
    for kv in filter(None, os.environ.get('SMX_BENCH_LEARNER_OPTS', '').split(',')):     # A/B runs: key=0|1[,key=...]
        k, v = kv.split('=')
        sc.learner[k] = bool(int(v))
    L = PPOLearner(lc, ppo_env_config(D, A), sc)
    batch = synthetic.make_ppo_batch(B, N, D, A, seed=1, rnn_hidden=HID)
    db = L._preprocess_batch_ppo(copy.deepcopy(batch))
    for _ in range(2): L.learn(db)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): st = L.learn(db)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print('RNN B=%4d N=%3d D=%3d A=%2d hid=%d H=%d: %.2f ms/learn  %.3g env-steps/s  (surr %.4f)' % (B, N, D, A, HID, horizon, dt * 1e3, B * N / dt, st['_surr_loss']))

if __name__ == '__main__':
    run(2, 25, 17, 6)
    run(64, 128, 17, 6)
    run(256, 128, 17, 6)
    if len(sys.argv) > 1:
        run(1024, 128, 376, 17, steps=2)
