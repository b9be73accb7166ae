"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path (surreal_amd/).

Loader that makes the *reference's own* hot-path modules importable in this
container so that golden vectors can be generated from the reference code
itself.  The reference (pure Python,
/root/reference) depends on third-party packages that are not installed and
cannot be installed here (torchx 0.9, caraml, tensorplex, benedict, gym,
imageio, cv2, symphony).  None of them carries hot-path arithmetic: their
layers are thin wrappers over torch ATen ops.  This module injects minimal
stand-ins into ``sys.modules``; the arithmetic that then runs is the
reference's own source (surreal/learner/ppo.py, surreal/model/ppo_net.py, ...).

/root/reference does not exist on the GPU box: only oracle/gen_golden.py (run
in the build container, output committed under tests/golden/) uses this file.

Unpinned by construction (torchx source absent): default weight init,
LayerNorm(1) semantics, LinearWithMinLR formula, state_dict key names.  Golden
vectors always carry the *injected* parameters so none of these matter for
parity.
"""
import collections
import collections.abc
import contextlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

# The reference's source tree exists in the build container only.  It does not travel to the GPU box in any form
# (no source, no bytecode): there `reference_available()` is False and every consumer falls back to the restatement.
REFERENCE_ROOT = os.environ.get('SURREAL_REFERENCE_ROOT') or '/root/reference'


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'surreal'))


# --------------------------------------------------------------------------
# benedict
# --------------------------------------------------------------------------
class BeneDict(dict):
    """attribute-access dict (recursive), enough for learner/base.py:10"""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, BeneDict):
            return BeneDict(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, BeneDict) else v)
                for k, v in self.items()}

    # yaml round trip used by surreal/utils/checkpoint.py:94-96, 255-256
    @classmethod
    def load_yaml_file(cls, path):
        import yaml
        with open(path) as fp:
            return cls(yaml.safe_load(fp))

    def dump_yaml_file(self, path):
        import yaml

        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [plain(x) for x in v]
            return v
        with open(path, 'w') as fp:
            yaml.safe_dump(plain(self), fp, default_flow_style=False)


# --------------------------------------------------------------------------
# torchx
# --------------------------------------------------------------------------
@contextlib.contextmanager
def _device_scope(*args, **kwargs):
    yield


class _TxModule(nn.Module):
    """torchx.nn.Module = nn.Module + a few helpers used by DDPG"""

    def hard_update(self, other):
        self.load_state_dict(other.state_dict())

    def soft_update(self, other, tau):
        for p, q in zip(self.parameters(), other.parameters()):
            p.data.copy_(p.data * (1.0 - tau) + q.data * tau)

    def clip_grad_value(self, clip):
        nn.utils.clip_grad_value_(self.parameters(), clip)

    def clip_grad_norm(self, clip):
        return nn.utils.clip_grad_norm_(self.parameters(), clip)


class LinearWithMinLR(object):
    """stand-in (formula unpinned): linear anneal to min_lr over num_updates"""

    def __init__(self, optimizer, num_updates, update_freq=1, min_lr=0.0):
        self.optimizer = optimizer
        self.num_updates = max(int(num_updates), 1)
        self.update_freq = update_freq
        self.min_lr = min_lr
        self.base_lrs = [g['lr'] for g in optimizer.param_groups]
        self.n = 0

    def get_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def step(self):
        self.n += 1
        if self.n % self.update_freq == 0:
            frac = max(0.0, 1.0 - self.n / self.num_updates)
            for g, b in zip(self.optimizer.param_groups, self.base_lrs):
                g['lr'] = max(self.min_lr, b * frac)

    def state_dict(self):
        return {'n': self.n}

    def load_state_dict(self, d):
        self.n = d['n']


class _Sym(object):
    def __init__(self, layer=None, parent=None, shape=None):
        self.layer, self.parent, self.shape = layer, parent, shape


class _Layer(nn.Module):
    """Keras-style: layer(sym) records the graph; .build(shape) creates params"""

    def __call__(self, x, *a, **kw):
        if isinstance(x, _Sym):
            return _Sym(self, x)
        return super().__call__(x, *a, **kw)

    def build_(self, in_shape):
        return in_shape

    def forward(self, x):
        return x


class Placeholder(_Sym):
    def __init__(self, shape):
        super().__init__(None, None, tuple(shape))


class Linear(_Layer):
    def __init__(self, out_features):
        super().__init__()
        self.out_features = out_features
        self.fc = None

    def build_(self, in_shape):
        self.fc = nn.Linear(in_shape[-1], self.out_features)
        return tuple(in_shape[:-1]) + (self.out_features,)

    def forward(self, x):
        return self.fc(x)


class ReLU(_Layer):
    def forward(self, x):
        return torch.relu(x)


class Tanh(_Layer):
    def forward(self, x):
        return torch.tanh(x)


class LayerNorm(_Layer):
    def __init__(self, ndims=1):
        super().__init__()
        self.ndims = ndims
        self.ln = None

    def build_(self, in_shape):
        self.ln = nn.LayerNorm(list(in_shape[-self.ndims:]))
        return in_shape

    def forward(self, x):
        return self.ln(x)


class Conv2d(_Layer):
    def __init__(self, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.oc, self.k, self.s, self.p = out_channels, kernel_size, stride, padding
        self.conv = None

    def build_(self, in_shape):
        c, h, w = in_shape[-3:]
        self.conv = nn.Conv2d(c, self.oc, self.k, self.s, self.p)
        oh = (h + 2 * self.p - self.k) // self.s + 1
        ow = (w + 2 * self.p - self.k) // self.s + 1
        return tuple(in_shape[:-3]) + (self.oc, oh, ow)

    def forward(self, x):
        return self.conv(x)


class Flatten(_Layer):
    def build_(self, in_shape):
        n = 1
        for d in in_shape[1:]:
            n *= d
        return (in_shape[0], n)

    def forward(self, x):
        return x.reshape(x.size(0), -1)


class Sequential(_Layer):
    def __init__(self, *layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def build(self, in_shape):
        s = tuple(in_shape)
        for l in self.layers:
            s = l.build_(s)
        return s

    build_ = build

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x


class Functional(_Layer):
    def __init__(self, inputs, outputs):
        super().__init__()
        chain, node = [], outputs
        while node is not inputs:
            chain.append(node.layer)
            node = node.parent
        self.layers = nn.ModuleList(chain[::-1])

    def build(self, in_shape):
        s = tuple(in_shape)
        for l in self.layers:
            s = l.build_(s)
        return s

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x


# --------------------------------------------------------------------------
class _Anything(object):
    """placeholder class for import-time-only names"""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """idempotent: inject the stand-in modules and put the reference on sys.path"""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError('reference tree not present at %s (it only exists in the '
                           'build container)' % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if not hasattr(collections, 'Sequence'):
        collections.Sequence = collections.abc.Sequence
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'float_'):
        np.float_ = np.float64
    if not hasattr(np, 'int'):
        np.int = int
    if not hasattr(np, 'bool'):
        np.bool = bool
    import torch.nn.init as tinit
    if not hasattr(tinit, 'xavier_uniform'):
        tinit.xavier_uniform = tinit.xavier_uniform_

    _mod('benedict', BeneDict=BeneDict)
    _mod('tensorplex', TensorplexClient=_Anything, LoggerplexClient=_Anything,
         Tensorplex=_Anything, Loggerplex=_Anything)
    zmq_names = ['ZmqSender', 'ZmqReceiver', 'ZmqProxyThread', 'ZmqPub', 'ZmqSub',
                 'ZmqServer', 'ZmqClient', 'DataFetcher', 'ZmqReq', 'ZmqPusher',
                 'ZmqPuller']
    zattrs = {n: _Anything for n in zmq_names}
    zattrs['ZmqTimeoutError'] = type('ZmqTimeoutError', (Exception,), {})
    cz = _mod('caraml.zmq', **zattrs)
    _mod('caraml', zmq=cz)

    class _Space(object):
        pass

    class Box(_Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete(_Space):
        def __init__(self, n):
            self.n = n

    spaces = _mod('gym.spaces', Box=Box, Discrete=Discrete, Space=_Space)

    class _GymEnv(object):
        pass

    class _GymWrapper(_GymEnv):
        def __init__(self, env=None):
            self.env = env

    _mod('gym', Env=_GymEnv, Wrapper=_GymWrapper, RewardWrapper=_GymWrapper,
         ObservationWrapper=_GymWrapper, ActionWrapper=_GymWrapper, spaces=spaces,
         make=_Anything())
    _mod('imageio')
    _mod('cv2')
    sym_addons = _mod('symphony.addons', DockerBuilder=_Anything)
    sym_cmd = _mod('symphony.commandline', SymphonyParser=_Anything)
    sym_eng = _mod('symphony.engine', SymphonyConfig=_Anything, Cluster=_Anything)
    _mod('symphony', addons=sym_addons, commandline=sym_cmd, engine=sym_eng)
    _mod('cloudwise')
    _mod('nanolog')

    hs = _mod('torchx.nn.hyper_scheduler', LinearWithMinLR=LinearWithMinLR,
              __all__=['LinearWithMinLR'])
    txnn = _mod('torchx.nn', Module=_TxModule, hyper_scheduler=hs)
    txl = _mod('torchx.layers', Placeholder=Placeholder, Linear=Linear, ReLU=ReLU,
               Tanh=Tanh, LayerNorm=LayerNorm, Conv2d=Conv2d, Flatten=Flatten,
               Sequential=Sequential, Functional=Functional)
    _mod('torchx', device_scope=_device_scope, nn=txnn, layers=txl)

    os.environ.setdefault('SYMPH_COLLECTOR_FRONTEND_HOST', '127.0.0.1')
    os.environ.setdefault('SYMPH_COLLECTOR_FRONTEND_PORT', '7000')
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def import_reference():
    """returns the reference ``surreal`` package's hot-path modules"""
    install()
    import surreal.learner.ppo as ppo
    import surreal.learner.aggregator as aggregator
    import surreal.model.ppo_net as ppo_net
    import surreal.model.z_filter as z_filter
    return types.SimpleNamespace(ppo=ppo, aggregator=aggregator, ppo_net=ppo_net,
                                 z_filter=z_filter)
