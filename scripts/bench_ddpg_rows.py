"""configs[2] shapes (batch 512, 17 -> 300 -> 200 -> 6 / 17 -> 400, [400 | 6] -> 300 -> 1): one DDPG iteration on the row-block
schedule against the level schedule, and -- with a library built with -DSMX_DDPG_TIMING
(python scripts/build_variant_lib.py ddpgt smx_ddpg_rows.hip -DSMX_DDPG_TIMING=1; SMX_LIB_PATH=...) -- the phases of the two
chain launches by cycle stamps."""
import copy, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from surreal_amd import synthetic
from surreal_amd.learner.ddpg import DDPGLearner
from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config

B, D, A = 512, 17, 6


def make(rows, graph=True):
    lc = ddpg_learner_config()
    lc.replay.batch_size = B
    sc = ddpg_session_config()
    sc.learner['ddpg_row_schedule'] = rows
    L = DDPGLearner(lc, ddpg_env_config(D, A), sc)
    L.use_graph = graph and L.use_graph
    return L


def dev_batch(seed):
    b = synthetic.make_ddpg_batch(B, D, A, seed=seed)
    def mv(x):
        if isinstance(x, dict):
            return {k: mv(v) for k, v in x.items()}
        return torch.as_tensor(x).cuda() if not torch.is_tensor(x) else x.cuda()
    return mv(b)


for rows in ((True,) if os.environ.get('SMX_ROWS_ONLY') else (True, False)):
    L = make(rows)
    bs = [dev_batch(s) for s in range(4)]
    for i in range(20):
        L.learn(copy.copy(bs[i % 4]))
    torch.cuda.synchronize()
    ts = []
    for w in range(3):
        t0 = time.perf_counter()
        for i in range(200):
            L.learn(copy.copy(bs[i % 4]))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200)
    print('%s schedule: %.4f ms per learn (median of 3 x 200)' % ('row-block' if rows else 'level', sorted(ts)[1] * 1e3))

if os.environ.get('SMX_DDPG_TBUF'):
    L = make(True, graph=False)
    lib = L.K.lib
    lib.smx_ddpg_rows_debug_tbuf.argtypes = [ctypes.c_void_p]
    lib.smx_ddpg_rows_debug_tbuf.restype = None
    b = dev_batch(0)
    for _ in range(3):
        L.learn(copy.copy(b))
    nb = (B + 3) // 4
    names_c = ['prologue+ta.L1', 'ta.L2', 'ta.L3', 'tc.L1', 'tc.L2', 'tc.L3', 'c.L1', 'c.L2', 'c.L3', 'loss+dz2+dz1', 'a.L1',
               'a.L2', 'a.L3']
    names_a = ['prologue+c.L1', 'c.L2', 'c.L3', 'dz2+W2Thi', 'tanh+aW3T', 'aW2T']
    ws = L._ws
    args = L._rows_args(ws, ws.s_obs, ws.s_next, ws.s_act, ws.s_rew, ws.s_done)
    for which, names in (('critic', names_c), ('actor', names_a)):
        tb = torch.zeros(nb, 128, dtype=torch.int64, device='cuda')
        lib.smx_ddpg_rows_debug_tbuf(ctypes.c_void_p(tb.data_ptr()))
        L.K.ddpg_rows_pack(args)
        getattr(L.K, 'ddpg_rows_' + which)(args)
        torch.cuda.synchronize()
        lib.smx_ddpg_rows_debug_tbuf(None)
        full = tb.cpu().numpy().astype(np.float64)
        t = full[:, :len(names) + 1]
        d = np.diff(t, axis=1)
        print('%s launch, cycles per phase (median over %d workgroups; total %.0f):' % (which, nb, np.median(t[:, -1] - t[:, 0])))
        for k, (n, v) in enumerate(zip(names, np.median(d, axis=0))):
            ds = np.median(np.diff(full[:, 16 + 5 * k:21 + 5 * k], axis=1), axis=0)
            print('   %-14s %8.0f   inside the layer (wave 0): entry %.0f  K loop %.0f  epilogue %.0f  barrier %.0f'
                  % (n, v, *ds))

if os.environ.get('SMX_DDPG_TBUF'):
    # the fused weight-gradient + update launch (critic group), by phase
    nbw = 160
    tb = torch.zeros(nbw, 128, dtype=torch.int64, device='cuda')
    lib.smx_ddpg_rows_debug_tbuf(ctypes.c_void_p(tb.data_ptr()))
    soft = L.target_update_type == 'soft'
    L.K.ddpg_rows_update(args, 'critic', L.model.critic_flat, ws.grads_c, L.critic_exp_avg, L.critic_exp_avg_sq, ws.lr[1:2], ws.step,
                         L.critic_regularization, L.critic_gradient_clip_value, target=L.model_target.critic_flat,
                         tau=L.target_update_tau if soft else 0.0, interval=0 if soft else L.target_update_interval, wgrad=True)
    torch.cuda.synchronize()
    lib.smx_ddpg_rows_debug_tbuf(None)
    t = tb.cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0][:, :6]
    d = np.diff(t, axis=1)
    print('wgrad + update launch (critic), %d workgroups, cycles (median): loads issued %.0f  params requested + bias corrections %.0f  '
          'products (waits for the loads) %.0f  meet + barrier %.0f  epilogue %.0f  total %.0f; first start -> last end %.0f'
          % ((len(t),) + tuple(np.median(d, axis=0)) + (np.median(t[:, 5] - t[:, 0]), t[:, 5].max() - t[:, 0].min())))
