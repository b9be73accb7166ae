"""micro-benchmark of smx_epoch_fwdbwd_f32 (one launch: forward + loss + data gradients) against the two launches it
replaces, at the benchmark shape (GPU box).  With a timing build (python scripts/build_timing_lib.py;
SMX_LIB_PATH=surreal_amd/libsurreal_amd_timing.so) also the per-phase cycle counts of thread 0 of every workgroup."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from surreal_amd import _lib as L
from surreal_amd.kernels import HipKernels
import test_gpu_epoch as TE
K = HipKernels()
mode = L.SMX_PPO_CLIP if 'clip' in sys.argv else L.SMX_PPO_ADAPT
rows, D, H1, H2, A = 1024, 376, 300, 200, 17
t = TE.build(rows, D, H1, H2, A, seed=1, mode=mode, device='cuda')['d']
t['ctrl'][L.C_KL_TARGET] = 1e9
NS = 4096
sync = torch.zeros(NS, dtype=torch.int32, device='cuda')
slots = torch.zeros(NS, 2 * ((rows + 15) // 16), dtype=torch.int32, device="cuda")
loss = dict(mode=mode, rows=rows, log_var=t['log_var'], actions=t['actions'], behave=t['behave'], ref=t['ref'],
            adv=t['adv'], g_surr=t['g_surr'], g_kl=t['g_kl'], partials=t['partials'], check_stop=True,
            will_update=True, dlogvar=t['dlogvar'], dlogvar_sumsq=t['dlq'], stats=t['stats'],
            returns=t['returns'], v_dz3=t['v_dz3'], v_partials=t['v_partials'], v_will_update=True)
K.epoch_pack([(t['act'], t['pk_a']), (t['cri'], t['pk_c'])])
aj = dict(net=t['act'], packed=t['pk_a'], x=t['x'], h1T=t['h1aT'], h2T=t['h2aT'], act=L.SMX_ACT_TANH, loss='policy',
          dz3T=t['dz3aT'], dz2T=t['dz2aT'], dz1T=t['dz1aT'], xT=t['xT'], grads=t['grads_a'])
cj = dict(net=t['cri'], packed=t['pk_c'], x=t['x'], h1T=t['h1cT'], h2T=t['h2cT'], act=L.SMX_ACT_NONE,
          loss='value', dz3=t['v_dz3'], dz3T=t['v_dz3'], dz2T=t['dz2cT'], dz1T=t['dz1cT'], xT=t['xT'], grads=t['grads_c'])
k = [0]


def two():
    K.epoch_forward([aj, cj], loss, t['ctrl'], rows)
    K.epoch_backward([aj, cj], loss, t['ctrl'], rows)


def one():
    K.epoch_fwdbwd([aj, cj], loss, t['ctrl'], rows, sync[k[0]:k[0] + 1], slots[k[0]])
    k[0] += 1


def graph_time(fn, n=50):
    """n calls captured in ONE hipGraph (what the learner replays): device time per call, no host in the way"""
    fn(); torch.cuda.synchronize()
    k0 = k[0]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e9
    for _ in range(5):
        sync.zero_(); slots.zero_(); t['ctrl'].view(torch.int32)[L.C_STEP_ACTOR:].zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print('two launches (forward, backward): %.2f us per epoch' % graph_time(two))
print('one launch (fwdbwd):              %.2f us per epoch' % graph_time(one))
if 'timing' in os.environ.get('SMX_LIB_PATH', ''):
    lib = K.lib
    lib.smx_epoch_debug_tbuf.argtypes = [ctypes.c_void_p]
    lib.smx_epoch_debug_tbuf.restype = None
    tb = torch.zeros(512 * 32, dtype=torch.int64, device='cuda')
    lib.smx_epoch_debug_tbuf(ctypes.c_void_p(tb.data_ptr()))
    sync.zero_(); slots.zero_(); k[0] = 0
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    NB = (rows + 15) // 16                     # row blocks per job (the launch: actor blocks, then critic blocks)
    TT = tb.view(512, 32)[:2 * NB].cpu().double()
    base = TT[:, 0].min()
    names = [(0, 12, 'job descriptor'), (12, 13, 'x + loss input loads issued'), (13, 14, 'hidden tiles cleared'),
             (14, 15, 'x -> LDS (waits for x)'), (15, 1, 'loss inputs -> LDS'), (0, 1, 'prologue'), (1, 2, 'barrier'), (2, 3, 'layer 1'), (3, 4, 'layer 2'), (4, 5, 'layer 3'),
             (5, 6, 'loss + publish barrier'), (6, 7, 'rhs tiles + dz2 products'), (7, 8, 'wait + reduce sums'),
             (8, 9, 'dz2 epilogue + dz3T + barrier'), (9, 10, 'dz1'), (10, 11, 'scalars')]
    for l, prev in ((0, 2), (1, 3), (2, 4)):
        names += [(prev, 16 + 4 * l, 'layer %d: set-up, bias loads issued' % (l + 1)), (16 + 4 * l, 17 + 4 * l, 'layer %d: products' % (l + 1)),
                  (17 + 4 * l, 18 + 4 * l, 'layer %d: epilogue' % (l + 1)), (18 + 4 * l, 3 + l, 'layer %d: barrier' % (l + 1))]
    for a, b, nm in names:
        d = TT[:, b] - TT[:, a]
        print('%-32s actor %7.0f (max %7.0f)   critic %7.0f (max %7.0f) cycles' % (
            nm, d[:NB].mean(), d[:NB].max(), d[NB:].mean(), d[NB:].max()))
    tot = TT[:, 11] - TT[:, 0]
    print('in-kernel total: actor %.0f (max %.0f) critic %.0f ; last end - first start %.0f cycles; start spread %.0f' % (
        tot[:NB].mean(), tot[:NB].max(), tot[NB:].mean(), TT[:, 11].max() - base, TT[:, 0].max() - base))
    lib.smx_epoch_debug_tbuf(None)
