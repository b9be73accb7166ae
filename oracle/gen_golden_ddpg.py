#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY.  tests/golden/ddpg_*.npz from the REFERENCE's own DDPGLearner
(surreal/learner/ddpg.py run under oracle/ref_shims.py, build container only), several
consecutive iterations with injected parameters; cross-checks oracle/ddpg_oracle.py bit-for-bit."""
import collections
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402
ref_shims.install()
from surreal_amd import synthetic  # noqa: E402
import ddpg_oracle  # noqa: E402
import surreal.utils as U  # noqa: E402
from surreal.learner.ddpg import DDPGLearner  # noqa: E402
from surreal.model.ddpg_net import DDPGModel  # noqa: E402


def lin_layers(net_functional):
    return [l for l in net_functional.layers if hasattr(l, 'fc')]


def ln_layers(net_functional):
    """the L.LayerNorm(1) layers (ref_shims.LayerNorm: torch.nn.LayerNorm over the last dimension) in network order"""
    return [l for l in net_functional.layers if hasattr(l, 'ln')]


def _ln_sets(m):
    out = []
    if m.actor is not None:
        out += [('actor.ln%d' % (i + 1), l) for i, l in enumerate(ln_layers(m.actor.model))]
    out += [('critic.ln%d' % (i + 1), l) for i, l in enumerate(ln_layers(m.critic.model_obs) + ln_layers(m.critic.model_concat))]
    return out


def _perception_layers(m):
    ls = [l for l in m.perception.model.layers if hasattr(l, 'conv') or hasattr(l, 'fc')]
    return collections.OrderedDict(zip(('conv1', 'conv2', 'fc'), ls))


def inject(m, params):
    with torch.no_grad():
        if getattr(m, 'is_pixel_input', False):
            for nm, layer in _perception_layers(m).items():
                mod = layer.conv if hasattr(layer, 'conv') else layer.fc
                mod.weight.copy_(torch.tensor(params['cnn.%s.W' % nm]))
                mod.bias.copy_(torch.tensor(params['cnn.%s.b' % nm]))
        for i, l in enumerate(lin_layers(m.actor.model) if m.actor is not None else []):
            l.fc.weight.copy_(torch.tensor(params['actor.fc%d.W' % (i + 1)]))
            l.fc.bias.copy_(torch.tensor(params['actor.fc%d.b' % (i + 1)]))
        c = lin_layers(m.critic.model_obs) + lin_layers(m.critic.model_concat)
        for i, l in enumerate(c):
            l.fc.weight.copy_(torch.tensor(params['critic.fc%d.W' % (i + 1)]))
            l.fc.bias.copy_(torch.tensor(params['critic.fc%d.b' % (i + 1)]))
        for name, l in _ln_sets(m):
            l.ln.weight.copy_(torch.tensor(params[name + '.W']))
            l.ln.bias.copy_(torch.tensor(params[name + '.b']))


def extract_perception(m):
    out = collections.OrderedDict()
    if getattr(m, 'is_pixel_input', False):
        for nm, layer in _perception_layers(m).items():
            mod = layer.conv if hasattr(layer, 'conv') else layer.fc
            out['cnn.%s.W' % nm] = mod.weight.detach().numpy().copy()
            out['cnn.%s.b' % nm] = mod.bias.detach().numpy().copy()
    return out


def extract(m):
    out = collections.OrderedDict()
    for i, l in enumerate(lin_layers(m.actor.model)):
        out['actor.fc%d.W' % (i + 1)] = l.fc.weight.detach().numpy().copy()
        out['actor.fc%d.b' % (i + 1)] = l.fc.bias.detach().numpy().copy()
    for i, l in enumerate(lin_layers(m.critic.model_obs) + lin_layers(m.critic.model_concat)):
        out['critic.fc%d.W' % (i + 1)] = l.fc.weight.detach().numpy().copy()
        out['critic.fc%d.b' % (i + 1)] = l.fc.bias.detach().numpy().copy()
    for name, l in _ln_sets(m):
        out[name + '.W'] = l.ln.weight.detach().numpy().copy()
        out[name + '.b'] = l.ln.bias.detach().numpy().copy()
    out.update(extract_perception(m))
    return out


def extract_critic(m):
    out = collections.OrderedDict()
    for i, l in enumerate(lin_layers(m.critic.model_obs) + lin_layers(m.critic.model_concat)):
        out['critic.fc%d.W' % (i + 1)] = l.fc.weight.detach().numpy().copy()
        out['critic.fc%d.b' % (i + 1)] = l.fc.bias.detach().numpy().copy()
    for name, l in _ln_sets(m):
        if name.startswith('critic.'):
            out[name + '.W'] = l.ln.weight.detach().numpy().copy()
            out[name + '.b'] = l.ln.bias.detach().numpy().copy()
    out.update(extract_perception(m))
    return out


class _Timer(object):
    avg = 0.0

    def time(self):
        import contextlib
        return contextlib.nullcontext()


def build_ref(params, D, A, ah, ch, hyper, params2=None, pixel=None, conv_hidden=200):
    L = object.__new__(DDPGLearner)
    L.batch_size = hyper['B']
    L.discount_factor, L.n_step = hyper['gamma'], hyper['n_step']
    L.is_pixel_input = pixel is not None
    L.use_double_critic = bool(hyper.get('double_critic', False))
    L.use_action_regularization = bool(hyper.get('action_reg', False))
    L.gpu_ids, L._num_gpus = 'cpu', 0
    L.clip_actor_gradient, L.actor_gradient_clip_value = True, 1.0
    L.clip_critic_gradient, L.critic_gradient_clip_value = hyper.get('clip_critic', False), 5.0
    L.action_dim = A
    obs_spec = collections.OrderedDict()
    if pixel is not None:                                   # ddpg_net.py:37-44: the CNN perception
        obs_spec['pixel'] = collections.OrderedDict(camera0=list(pixel))
    obs_spec['low_dim'] = collections.OrderedDict(flat_inputs=[D])
    mk = lambda **kw: DDPGModel(obs_spec=obs_spec, action_dim=A, use_layernorm=bool(hyper.get('layernorm', False)),  # noqa: E731
                                actor_fc_hidden_sizes=list(ah), critic_fc_hidden_sizes=list(ch),
                                conv_out_channels=[16, 32], conv_kernel_sizes=[8, 4], conv_strides=[4, 2],
                                conv_hidden_dim=conv_hidden, **kw)
    L.model, L.model_target = mk(), mk()
    inject(L.model, params)
    inject(L.model_target, params)
    L.critic_criterion = torch.nn.MSELoss()
    L.critic_optim = torch.optim.Adam(L.model.get_critic_parameters(), lr=hyper['lr_critic'])
    if L.use_double_critic:                                # ddpg.py:119-147, 162-166
        L.model2, L.model_target2 = mk(critic_only=True), mk(critic_only=True)
        inject(L.model2, params2)
        inject(L.model_target2, params2)
        L.critic_optim2 = torch.optim.Adam(L.model2.get_critic_parameters(), lr=hyper['lr_critic'])
    L.actor_optim = torch.optim.Adam(L.model.get_actor_parameters(), lr=hyper['lr_actor'])
    L.target_update_type = hyper['target_update_type']
    L.target_update_counter = 0
    L.target_update_interval = hyper['target_update_interval']
    L.target_update_tau = hyper.get('tau', 1e-3)
    L.forward_time = L.critic_update_time = L.actor_update_time = _Timer()
    return L


CASES = {
    'tiny_hard': dict(B=16, D=5, A=2, ah=(24, 16), ch=(32, 24), iters=4,
                      hyper=dict(gamma=0.99, n_step=3, lr_actor=1e-3, lr_critic=1e-2,
                                 target_update_type='hard', target_update_interval=2)),
    'tiny_soft_clipcritic': dict(B=37, D=9, A=3, ah=(40, 24), ch=(48, 40), iters=3,
                                 hyper=dict(gamma=0.9, n_step=1, lr_actor=1e-3, lr_critic=1e-2,
                                            target_update_type='soft', target_update_interval=1,
                                            tau=0.05, clip_critic=True)),
    'tiny_td3_hard': dict(B=16, D=5, A=2, ah=(24, 16), ch=(32, 24), iters=4,
                          hyper=dict(gamma=0.99, n_step=3, lr_actor=1e-3, lr_critic=1e-2, double_critic=True,
                                     action_reg=True, target_update_type='hard', target_update_interval=2)),
    'tiny_double_soft': dict(B=21, D=7, A=3, ah=(24, 16), ch=(32, 24), iters=3,
                             hyper=dict(gamma=0.95, n_step=2, lr_actor=1e-3, lr_critic=1e-2, double_critic=True,
                                        target_update_type='soft', target_update_interval=1, tau=0.1,
                                        clip_critic=True)),
    'tiny_pixel_hard': dict(B=12, D=4, A=2, ah=(24, 16), ch=(32, 24), iters=4, pixel=(2, 20, 24), conv_hidden=8,
                            hyper=dict(gamma=0.99, n_step=3, lr_actor=1e-3, lr_critic=1e-2,
                                       target_update_type='hard', target_update_interval=2)),
    'tiny_pixel_td3_soft': dict(B=10, D=3, A=2, ah=(24, 16), ch=(32, 24), iters=3, pixel=(3, 28, 36), conv_hidden=16,
                                hyper=dict(gamma=0.95, n_step=2, lr_actor=1e-3, lr_critic=1e-2, double_critic=True,
                                           action_reg=True, target_update_type='soft', target_update_interval=1,
                                           tau=0.1, clip_critic=True)),
    # use_layernorm = True (reference default: off): a LayerNorm behind every hidden ReLU of both networks
    'tiny_ln_hard': dict(B=16, D=5, A=2, ah=(24, 16), ch=(32, 24), iters=4,
                         hyper=dict(gamma=0.99, n_step=3, lr_actor=1e-3, lr_critic=1e-2, layernorm=True,
                                    target_update_type='hard', target_update_interval=2)),
    'ln_soft_clipcritic': dict(B=37, D=9, A=3, ah=(72, 40), ch=(80, 72), iters=3,
                               hyper=dict(gamma=0.9, n_step=1, lr_actor=1e-3, lr_critic=1e-2, layernorm=True,
                                          target_update_type='soft', target_update_interval=1,
                                          tau=0.05, clip_critic=True)),
    # round 6: use_layernorm together with the other switches (builders.py:42-48, 65-74: the flag reaches every network,
    # the second critic and the networks on top of the perception CNN included)
    'tiny_ln_td3_soft': dict(B=21, D=7, A=3, ah=(24, 16), ch=(32, 24), iters=3,
                             hyper=dict(gamma=0.95, n_step=2, lr_actor=1e-3, lr_critic=1e-2, layernorm=True,
                                        double_critic=True, action_reg=True, target_update_type='soft',
                                        target_update_interval=1, tau=0.1, clip_critic=True)),
    'tiny_ln_pixel_hard': dict(B=12, D=4, A=2, ah=(24, 16), ch=(32, 24), iters=4, pixel=(2, 20, 24), conv_hidden=8,
                               hyper=dict(gamma=0.99, n_step=3, lr_actor=1e-3, lr_critic=1e-2, layernorm=True,
                                          target_update_type='hard', target_update_interval=2)),
    'tiny_ln_pixel_td3_soft': dict(B=10, D=3, A=2, ah=(24, 16), ch=(32, 24), iters=3, pixel=(3, 28, 36), conv_hidden=16,
                                   hyper=dict(gamma=0.95, n_step=2, lr_actor=1e-3, lr_critic=1e-2, layernorm=True,
                                              double_critic=True, action_reg=True, target_update_type='soft',
                                              target_update_interval=1, tau=0.1, clip_critic=True)),
    'cfg3_cheetah512': dict(B=512, D=17, A=6, ah=(300, 200), ch=(400, 300), iters=3,
                            hyper=dict(gamma=0.99, n_step=3, lr_actor=1e-4, lr_critic=1e-3,
                                       target_update_type='hard', target_update_interval=500)),
    # round 6 (VERDICT r05 "missing" 4): configs[2]'s size ACROSS the hard target update of ddpg.py:403-428 at the
    # reference's interval (main/ddpg_configs.py:54-60): 502 iterations, so the target is the model of iteration 500 and
    # the model has moved twice since; every iteration's statistics, final and target parameters are stored
    'cfg3_cheetah512_x502': dict(B=512, D=17, A=6, ah=(300, 200), ch=(400, 300), iters=502,
                                 hyper=dict(gamma=0.99, n_step=3, lr_actor=1e-4, lr_critic=1e-3,
                                            target_update_type='hard', target_update_interval=500)),
}


def main(only=None):
    for name, c in CASES.items():
        if only and name not in only:
            continue
        hyper = dict(c['hyper'], B=c['B'])
        pixel = tuple(c['pixel']) if c.get('pixel') else None
        if pixel is not None:
            mkp = lambda seed: ddpg_oracle.make_ddpg_pixel_params(  # noqa: E731
                c['D'], c['A'], pixel, c['conv_hidden'], c['ah'], c['ch'], seed=seed,
                layernorm=bool(hyper.get('layernorm', False)))
        else:
            mkp = lambda seed: ddpg_oracle.make_ddpg_params(c['D'], c['A'], c['ah'], c['ch'], seed=seed,  # noqa: E731
                                                            layernorm=bool(hyper.get('layernorm', False)))
        params, params2 = mkp(3), mkp(4)
        Lr = build_ref(params, c['D'], c['A'], c['ah'], c['ch'], hyper, params2, pixel=pixel,
                       conv_hidden=c.get('conv_hidden', 200))
        O = ddpg_oracle.OracleDDPGLearner(
            use_double_critic=hyper.get('double_critic', False),
            use_action_regularization=hyper.get('action_reg', False), params2=params2, batch_size=c['B'],
            params=params, gamma=hyper['gamma'], n_step=hyper['n_step'], lr_actor=hyper['lr_actor'],
            lr_critic=hyper['lr_critic'], clip_critic_gradient=hyper.get('clip_critic', False),
            target_update_type=hyper['target_update_type'],
            target_update_interval=hyper['target_update_interval'], tau=hyper.get('tau', 1e-3))
        traces = []
        for it in range(c['iters']):
            b = synthetic.make_ddpg_batch(c['B'], c['D'], c['A'], seed=10 + it, pixel=pixel)
            t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)  # noqa: E731

            def conv(o):                     # what DDPGLearner.preprocess hands over (ddpg.py:207-222)
                out = collections.OrderedDict()
                if pixel is not None:
                    out['pixel'] = {'camera0': torch.tensor(o['pixel']['camera0'], dtype=torch.uint8).float()}
                out['low_dim'] = {'flat_inputs': t(o['low_dim']['flat_inputs'])}
                return out
            obs, obs_next = conv(b['obs']), conv(b['obs_next'])
            np.random.seed(1000 + it)          # the action-regularisation noise comes from numpy's stream
            sr = Lr._optimize(obs, t(b['actions']), t(b['rewards']), obs_next, t(b['dones']))
            sr = {k: float(v) for k, v in sr.items() if not k.startswith('performance')}
            np.random.seed(1000 + it)
            so = O.learn(b)
            for k in sr:
                assert sr[k] == so[k], (name, it, k, sr[k], so[k])
            traces.append(sr)
        fr, fo = extract(Lr.model), O.model.numpy_params()
        for k in fr:
            np.testing.assert_array_equal(fr[k], fo[k])
        ft = extract(Lr.model_target)
        for k in ft:
            np.testing.assert_array_equal(ft[k], O.model_target.numpy_params()[k])
        if hyper.get('double_critic'):
            f2 = {k: v for k, v in extract_critic(Lr.model2).items()}
            for k, v in f2.items():
                np.testing.assert_array_equal(v, O.model2.numpy_params()[k])
            t2 = extract_critic(Lr.model_target2)
            for k, v in t2.items():
                np.testing.assert_array_equal(v, O.model_target2.numpy_params()[k])
        out = {'case_json': np.array(json.dumps({k: (list(v) if isinstance(v, tuple) else v)
                                                 for k, v in c.items()})),
               'trace_json': np.array(json.dumps(traces))}
        if True:                             # (round 6: the 512-row cases store their parameters too)
            for k, v in fr.items():
                out['final.' + k] = v
            for k, v in ft.items():
                out['target.' + k] = v
        if hyper.get('double_critic'):
            for k, v in f2.items():
                out['final2.' + k] = v
            for k, v in t2.items():
                out['target2.' + k] = v
        out['final_sumsq_json'] = np.array(json.dumps(
            {k: float(np.sum(v.astype(np.float64) ** 2)) for k, v in fr.items()}))
        path = os.path.join(ROOT, 'tests', 'golden', 'ddpg_%s.npz' % name)
        np.savez_compressed(path, **out)
        print(name, traces[-1], os.path.getsize(path))


if __name__ == '__main__':
    main(sys.argv[1:] or None)
