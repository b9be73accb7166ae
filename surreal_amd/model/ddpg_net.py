"""
DDPGModel on MI355X (surreal/model/ddpg_net.py:13-95; ActorNetworkX / CriticNetworkX,
model_builders/builders.py:35-84).  use_layernorm=True (reference default: off, ddpg_configs.py:21) puts a LayerNorm over
the features behind every hidden ReLU (builders.py:42-48, 65-75; torchx's L.LayerNorm(1) taken as torch.nn.LayerNorm(F) -- its
source is not in the reference tree) -- low-dimensional observations only.  With camera frames in the observation the "perception" CNN
(CNNStemNetwork, builders.py:8-33; ddpg_net.py:37-44, 67-78) runs on ``camera0 / 255`` and its
features are concatenated IN FRONT of the low-dim vector; it trains with the critic
(ddpg_net.py:57-61).

  actor : Linear(D,h1)-ReLU[-LN]-Linear(h1,h2)-ReLU[-LN]-Linear(h2,A)-Tanh
  critic: Linear(D,c1)-ReLU[-LN] ; concat action ; Linear(c1+A,c2)-ReLU[-LN] ; Linear(c2,1)

One flat fp32 parameter buffer per network (actor / critic / perception).
"""
import collections

import numpy as np
import torch

from surreal_amd import kernels as KN
from surreal_amd.model.cnn_stem import CnnParams, CnnStem
from surreal_amd.model.ppo_net import Mlp3Params


class DDPGModel(object):
    def __init__(self, obs_spec, action_dim, use_layernorm, actor_fc_hidden_sizes,
                 critic_fc_hidden_sizes, conv_out_channels=None, conv_kernel_sizes=None,
                 conv_strides=None, conv_hidden_dim=None, critic_only=False, device=None,
                 kernels=None):
        self.K = kernels or KN.default_kernels()
        device = device or KN.default_device()
        self._ctor = dict(obs_spec=obs_spec, action_dim=action_dim, use_layernorm=use_layernorm,
                          actor_fc_hidden_sizes=list(actor_fc_hidden_sizes),
                          critic_fc_hidden_sizes=list(critic_fc_hidden_sizes),
                          conv_out_channels=conv_out_channels, conv_kernel_sizes=conv_kernel_sizes,
                          conv_strides=conv_strides, conv_hidden_dim=conv_hidden_dim,
                          critic_only=critic_only, device=device, kernels=self.K)
        self.is_pixel_input = 'pixel' in obs_spec
        self.action_dim = A = action_dim
        self.use_layernorm = use_layernorm
        self.low_dim = int(obs_spec['low_dim']['flat_inputs'][0]) if 'low_dim' in obs_spec else 0
        self.cnn = self.perception_flat = self._cnn_stem = None
        self.feat_dim = 0
        if self.is_pixel_input:
            if len(obs_spec['pixel']) != 1 or 'camera0' not in obs_spec['pixel']:
                raise NotImplementedError('one camera (camera0), as ddpg_net.py:34-35')
            cam = tuple(int(v) for v in obs_spec['pixel']['camera0'])
            self.feat_dim = int(conv_hidden_dim)
            geo = dict(conv_channels=tuple(conv_out_channels or (16, 32)),
                       kernel_sizes=tuple(conv_kernel_sizes or (8, 4)), strides=tuple(conv_strides or (4, 2)))
            self.perception_flat = torch.empty(CnnParams.count(cam, self.feat_dim, **geo), device=device)
            self.cnn = CnnParams(self.perception_flat, 0, cam, self.feat_dim, **geo)
            self._cnn_stem = CnnStem(self.K)
        self.input_dim = D = self.feat_dim + self.low_dim          # ddpg_net.py:36-46
        self.device = device
        ah, ch = list(actor_fc_hidden_sizes), list(critic_fc_hidden_sizes)
        self.c1, self.c2 = ch
        c1, c2 = ch
        sizes = [('W1', (c1, D)), ('b1', (c1,)), ('W2', (c2, c1 + A)), ('b2', (c2,)),
                 ('W3', (1, c2)), ('b3', (1,))]
        if use_layernorm:            # the affine parameters of the two LayerNorms, behind the dense ones
            sizes += [('ln1.W', (c1,)), ('ln1.b', (c1,)), ('ln2.W', (c2,)), ('ln2.b', (c2,))]
        n = sum(int(np.prod(s)) for _, s in sizes)
        self.ln_eps = 1e-5           # torch.nn.LayerNorm's default
        if use_layernorm and max(list(actor_fc_hidden_sizes) + list(critic_fc_hidden_sizes)) > 1024:
            raise NotImplementedError('use_layernorm: hidden sizes up to 1024 (smx_layernorm_*_f32 keep a row in one workgroup)')
        self.actor_flat = self.ac_flat = None
        if not critic_only:
            # actor and critic parameters live in ONE buffer (each on a 256-byte boundary, the gap zero): the target
            # network's soft / hard update is then one launch over `ac_flat` instead of one per network
            n_mlp = Mlp3Params.count(D, ah[0], ah[1], A)
            na = n_mlp + (2 * (ah[0] + ah[1]) if use_layernorm else 0)
            na_pad = (na + 63) // 64 * 64
            self.ac_flat = torch.zeros(na_pad + n, device=device)
            self.actor_flat = self.ac_flat[:na]
            self.actor = Mlp3Params(self.actor_flat, 0, D, ah[0], ah[1], A)
            self.actor_ln = collections.OrderedDict()
            if use_layernorm:
                o = n_mlp
                for name, k in (('ln1.W', ah[0]), ('ln1.b', ah[0]), ('ln2.W', ah[1]), ('ln2.b', ah[1])):
                    self.actor_ln[name] = self.actor_flat[o:o + k]
                    o += k
            self.critic_flat = self.ac_flat[na_pad:na_pad + n]
        else:
            self.actor = None
            self.critic_flat = torch.empty(n, device=device)
        self.critic = collections.OrderedDict()
        o = 0
        for name, shp in sizes:
            k = int(np.prod(shp))
            self.critic[name] = self.critic_flat[o:o + k].view(*shp)
            o += k
        self._init()

    def _init(self):
        if self.cnn is not None:
            self.perception_flat.zero_()
            self.cnn.init_torch_default()
        nets = [self.critic] + ([self.actor.views] if self.actor is not None else [])
        for views in nets:
            for name, v in views.items():
                if name.startswith('ln'):
                    continue
                fan = views['W' + name[1]].shape[1]
                v.uniform_(-1.0 / np.sqrt(fan), 1.0 / np.sqrt(fan))
        if self.use_layernorm:           # torch.nn.LayerNorm: weight 1, bias 0
            for views in [self.critic] + ([self.actor_ln] if self.actor is not None else []):
                for name, v in views.items():
                    if name.startswith('ln'):
                        v.fill_(1.0 if name.endswith('.W') else 0.0)

    def named_parameters(self):
        out = collections.OrderedDict()
        if self.actor is not None:
            for i in (1, 2, 3):
                out['actor.fc%d.W' % i] = self.actor.views['W%d' % i]
                out['actor.fc%d.b' % i] = self.actor.views['b%d' % i]
        for i in (1, 2, 3):
            out['critic.fc%d.W' % i] = self.critic['W%d' % i]
            out['critic.fc%d.b' % i] = self.critic['b%d' % i]
        if self.use_layernorm:
            if self.actor is not None:
                for k, v in self.actor_ln.items():
                    out['actor.' + k] = v
            for k in ('ln1.W', 'ln1.b', 'ln2.W', 'ln2.b'):
                out['critic.' + k] = self.critic[k]
        if self.cnn is not None:
            for k, v in self.cnn.views.items():
                out['cnn.' + k] = v
        return out

    def load_params(self, params):
        with torch.no_grad():
            for k, v in self.named_parameters().items():
                v.copy_(torch.as_tensor(np.asarray(params[k]), dtype=torch.float32).view(v.shape))

    def numpy_params(self):
        return collections.OrderedDict((k, v.detach().cpu().numpy().copy())
                                       for k, v in self.named_parameters().items())

    def state_dict(self):
        return collections.OrderedDict(self.named_parameters())

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, v in self.named_parameters().items():
                src = sd[k]
                if not torch.is_tensor(src):
                    src = torch.as_tensor(np.asarray(src), dtype=torch.float32)
                v.copy_(src.view(v.shape))

    def get_actor_parameters(self):
        return [self.actor_flat]

    def get_critic_parameters(self):
        return [self.critic_flat] + ([self.perception_flat] if self.cnn is not None else [])

    # ---- forward passes (eager helpers; the learner uses its own workspace) -----------------
    def perception_into(self, frames, low, cnn_ws, out):
        """out [rows, feat + low_dim] = [CNN(frames / 255) | low]; cnn_ws keeps the activations (a
        backward pass may follow)"""
        rows = out.shape[0]
        self._cnn_stem.forward(self.cnn, frames, rows, cnn_ws, out[:, :self.feat_dim])
        if self.low_dim:
            out[:, self.feat_dim:].copy_(low)

    def forward_perception(self, obs):
        if not self.is_pixel_input:
            return obs['low_dim']['flat_inputs']
        frames = obs['pixel']['camera0'].contiguous()
        rows = frames.shape[0]
        out = torch.empty(rows, self.input_dim, device=frames.device)
        cws = CnnStem.workspace(self.cnn, rows, frames.device, backward=False)
        self.perception_into(frames, obs['low_dim']['flat_inputs'] if self.low_dim else None, cws, out)
        return out

    # ---- the LayerNorm variant: layer by layer, the LayerNorm inputs and row statistics kept for a backward pass ----
    def _ln_workspace_cached(self, rows, device):
        """the per-step act path (forward_actor / forward_critic): one workspace per (rows, device), not ~14 allocations
        per call"""
        key = (int(rows), str(device))
        cache = self.__dict__.setdefault('_ln_ws_cache', {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = self.ln_workspace(rows, device)
        return cache[key]

    def ln_workspace(self, rows, device=None):
        """buffers of one actor and one critic pass with LayerNorm (activations in front of each LayerNorm, its outputs,
        row means / reciprocal standard deviations)"""
        import types
        f = lambda *s: torch.empty(*s, device=device or self.device)  # noqa: E731
        w = types.SimpleNamespace()
        if self.actor is not None:
            a = self.actor
            w.a1, w.n1, w.a2, w.n2 = f(rows, a.H1), f(rows, a.H1), f(rows, a.H2), f(rows, a.H2)
            w.am1, w.ar1, w.am2, w.ar2 = f(rows), f(rows), f(rows), f(rows)
        w.c_a1, w.c_a2, w.c_n2 = f(rows, self.c1), f(rows, self.c2), f(rows, self.c2)
        w.cm1, w.cr1, w.cm2, w.cr2 = f(rows), f(rows), f(rows), f(rows)
        return w

    def actor_forward_ln(self, x, w, out):
        """out = tanh(fc3(LN(relu(fc2(LN(relu(fc1 x)))))))   (builders.py:35-56 with use_layernorm)"""
        K, a, v, ln = self.K, self.actor, self.actor.views, self.actor_ln
        rows, D = x.shape
        K.linear(x, 1, v['W1'], 1, v['b1'], w.a1, rows, a.H1, D, act=1)
        K.layernorm_forward(w.a1, ln['ln1.W'], ln['ln1.b'], self.ln_eps, w.n1, w.am1, w.ar1)
        K.linear(w.n1, 1, v['W2'], 1, v['b2'], w.a2, rows, a.H2, a.H1, act=1)
        K.layernorm_forward(w.a2, ln['ln2.W'], ln['ln2.b'], self.ln_eps, w.n2, w.am2, w.ar2)
        K.linear(w.n2, 1, v['W3'], 1, v['b3'], out, rows, a.OUT, a.H2, act=2)

    def critic_forward_ln(self, x, action, w, xcat, q):
        """xcat [rows, c1 + A] receives LN(relu(fc1 x)) | action; q [rows]   (builders.py:58-84 with use_layernorm)"""
        K, c = self.K, self.critic
        rows, D = x.shape
        A, c1, c2 = self.action_dim, self.c1, self.c2
        K.linear(x, 1, c['W1'], 1, c['b1'], w.c_a1, rows, c1, D, act=1)
        K.layernorm_forward(w.c_a1, c['ln1.W'], c['ln1.b'], self.ln_eps, xcat[:, :c1], w.cm1, w.cr1)
        xcat[:, c1:].copy_(action)
        K.linear(xcat, 1, c['W2'], 1, c['b2'], w.c_a2, rows, c2, c1 + A, act=1)
        K.layernorm_forward(w.c_a2, c['ln2.W'], c['ln2.b'], self.ln_eps, w.c_n2, w.cm2, w.cr2)
        K.linear(w.c_n2, 1, c['W3'], 1, c['b3'], q.view(rows, 1), rows, 1, c2, act=0)

    def forward_actor(self, x):
        a = self.actor
        rows = x.shape[0]
        if self.use_layernorm:
            out = torch.empty(rows, a.OUT, device=x.device)
            self.actor_forward_ln(x.contiguous(), self._ln_workspace_cached(rows, x.device), out)
            return out
        h1 = torch.empty(rows, a.H1, device=x.device)
        h2 = torch.empty(rows, a.H2, device=x.device)
        out = torch.empty(rows, a.OUT, device=x.device)
        self.K.mlp3_forward(a, x.contiguous(), h1, h2, out, 2)
        return out

    def critic_forward_into(self, x, action, xcat, h2, q):
        """xcat [rows, c1+A] receives relu(layer1) | action; h2 [rows, c2]; q [rows]"""
        K, c = self.K, self.critic
        rows, D = x.shape
        A, c1, c2 = self.action_dim, self.c1, self.c2
        K.linear(x, 1, c['W1'], 1, c['b1'], xcat, rows, c1, D, act=1, ldc=c1 + A)
        xcat[:, c1:].copy_(action)
        K.linear(xcat, 1, c['W2'], 1, c['b2'], h2, rows, c2, c1 + A, act=1)
        K.linear(h2, 1, c['W3'], 1, c['b3'], q.view(rows, 1), rows, 1, c2, act=0)

    def forward_critic(self, x, action):
        rows = x.shape[0]
        if self.use_layernorm:
            xcat = torch.empty(rows, self.c1 + self.action_dim, device=x.device)
            q = torch.empty(rows, device=x.device)
            self.critic_forward_ln(x.contiguous(), action, self._ln_workspace_cached(rows, x.device), xcat, q)
            return q.view(rows, 1)
        xcat = torch.empty(rows, self.c1 + self.action_dim, device=x.device)
        h2 = torch.empty(rows, self.c2, device=x.device)
        q = torch.empty(rows, device=x.device)
        self.critic_forward_into(x.contiguous(), action, xcat, h2, q)
        return q.view(rows, 1)

    def __deepcopy__(self, memo):
        """same architecture, own parameter buffers holding the same values (the kernel facade and
        the device are shared, not copied)"""
        twin = DDPGModel(**self._ctor)
        twin.load_state_dict(self.state_dict())
        return twin

    def __call__(self, obs_in, calculate_value=True, action=None):
        return self.forward(obs_in, calculate_value=calculate_value, action=action)

    def forward(self, obs_in, calculate_value=True, action=None):
        x = self.forward_perception(obs_in)
        if action is None:
            action = self.forward_actor(x)
        value = self.forward_critic(x, action) if calculate_value else None
        return action, value
