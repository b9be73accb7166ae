"""Which gradient tensors of the HIP path are further from exact arithmetic than the reference's fp32 path (GPU box)?
One policy and one value update at FIXED parameters (both learning rates 0): the gradient of every parameter tensor
from (a) the HIP path, (b) the oracle in fp32 on this host (= the reference's ATen ops), (c) the oracle in float64.
Prints, per tensor and optimiser group, |g - g64| / |g64| for (a) and (b).
    python tests/diag/diag_grad_accuracy.py cfg4_pixel_rnn_256x32"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch
import helpers as H
import ppo_oracle

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4_pixel_rnn_256x32'
if os.environ.get('SMX_DIAG_CPU'):             # dry run of this script on the torch-CPU kernel double
    from surreal_amd import kernels as KN
    from cpu_kernels import TorchCpuKernels
    KN.set_default_kernels(TorchCpuKernels(), 'cpu')
if name.startswith('seed'):             # a case of tests/golden/fp64_arbiter_seeds.json: python ... seed14
    case = json.load(open(H.SEEDS_PATH))['seeds'][name[4:]]['case']
else:
    g, case = H.load_golden(name)
case = copy.deepcopy(case)
case['hyper'].update(epoch_policy=1, epoch_baseline=1, lr_actor=0.0, lr_critic=0.0)
batch, params, zstate = H.case_inputs(case)


def oracle_grads(dtype):
    torch.set_default_dtype(dtype)
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
    got = {}
    for which, names in (('_policy_update', O.model.actor_params), ('_value_update', O.model.critic_params)):
        inner = getattr(O, which)

        def wrapped(*a, _inner=inner, _which=which, **k):
            out = _inner(*a, **k)
            got[_which] = {n: p.grad.detach().double().numpy().copy() for n, p in O.model.p.items() if p.grad is not None}
            return out
        setattr(O, which, wrapped)
    O.learn(copy.deepcopy(batch))
    torch.set_default_dtype(torch.float32)
    return got, O.trace


g64, t64 = oracle_grads(torch.float64)
g32, t32 = oracle_grads(torch.float32)
learner = H.make_learner(case, params, zstate, session_overrides={'use_hip_graph': False})
learner.learn(copy.deepcopy(batch))
ws, m = learner._ws, learner.model
flat0 = m.flat.data_ptr()
hip = {'_policy_update': {}, '_value_update': {}}
ga, gc = ws.grads_a.detach().cpu().double().numpy(), ws.grads_c.detach().cpu().double().numpy()
for n, v in m.named_parameters().items():
    off = (v.data_ptr() - flat0) // 4
    cnt = v.numel()
    if off + cnt <= ga.size and not n.startswith('critic.'):
        hip['_policy_update'][n] = ga[off:off + cnt].reshape(tuple(v.shape))
    co = off - m.n_actor_block
    if co >= 0:
        hip['_value_update'][n] = gc[co:co + cnt].reshape(tuple(v.shape))


def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-300))


out = {}
for which in ('_policy_update', '_value_update'):
    print('---- %s: |g - g64| / |g64|      HIP        ATen-fp32 (this host)     |g64|' % which)
    tot = {'hip': 0.0, 'f32': 0.0, 'n': 0.0}
    for n, ref in g64[which].items():
        if n not in hip[which]:
            continue
        a, b = hip[which][n].reshape(ref.shape), g32[which][n]
        ra, rb = rel(a, ref), rel(b, ref)
        out['%s %s' % (which, n)] = (ra, rb)
        tot['hip'] += np.sum((a - ref) ** 2); tot['f32'] += np.sum((b - ref) ** 2); tot['n'] += np.sum(ref ** 2)
        print('%-24s %10.2e %10.2e %s %12.4g' % (n, ra, rb, '  <--' if ra > 3 * rb and ra > 1e-6 else '     ', np.linalg.norm(ref)))
    print('%-24s %10.2e %10.2e   (whole group)' % ('group', np.sqrt(tot['hip'] / tot['n']), np.sqrt(tot['f32'] / tot['n'])))
for which, key in (('policy', 'grad_norm_actor'), ('value', 'grad_norm_critic')):
    print(key, 'hip %.9g  fp32 %.9g  fp64 %.9g' % (learner.trace[which][0][key], t32[which][0][key], t64[which][0][key]))
d = os.path.join(ROOT, 'gpurun_out')
if os.path.isdir(d):
    json.dump(out, open(os.path.join(d, 'grad_accuracy_%s.json' % name), 'w'), indent=0)

# ---- ReLU-mask lottery: how many of the CNN stem's output activations have a different sign in each fp32 path than
# in float64 (a unit within fp32 rounding of zero lands on either side; its whole gradient contribution flips) -----
if m.if_pixel:
    E, B = ws.E, ws.key[0]
    pix = np.asarray(batch['obs']['pixel']['camera0'])[:, :E]
    pix = pix.reshape((-1,) + pix.shape[2:])
    feats = {}
    for nm, dt in (('f64', torch.float64), ('f32', torch.float32)):
        torch.set_default_dtype(dt)
        M = ppo_oracle.OraclePPOModel(params, case['shape']['A'], True, zstate, in_size=case['shape']['D'])
        with torch.no_grad():
            feats[nm] = torch.cat([M._cnn(torch.as_tensor(pix[i:i + 512], dtype=dt) / 255.0) for i in range(0, len(pix), 512)]).double().numpy()
    torch.set_default_dtype(torch.float32)
    D = ws.key[2]
    feats['hip'] = ws.xn[:, D:].detach().cpu().double().numpy()
    for nm in ('hip', 'f32'):
        flips = np.argwhere((feats[nm] > 0) != (feats['f64'] > 0))
        print('CNN feature units whose ReLU mask differs from float64: %-4s %d of %d; max |feature - f64| %.2e' % (
            nm, len(flips), feats['f64'].size, np.abs(feats[nm] - feats['f64']).max()))
        for r, c in flips[:5]:
            print('    row %d unit %d: %s %.3e  f64 %.3e' % (r, c, nm, feats[nm][r, c], feats['f64'][r, c]))
