"""
PPO model on MI355X: DiagGauss, ZFilter and PPOModel with the reference's interface
(surreal/model/ppo_net.py:13-375, surreal/model/z_filter.py:7-107,
surreal/model/model_builders/builders.py:86-175) over HIP kernels.

Parameters live in HBM as ONE flat fp32 buffer per optimiser group
(actor: W1|b1|W2|b2|W3|b3|log_var, critic: W1|b1|W2|b2|W3|b3) so that clip_grad_norm_ + Adam
is a single launch and a data-parallel gradient all-reduce is a single collective.
"""
import collections
import ctypes

import numpy as np
import torch

from surreal_amd import _lib as L
from surreal_amd import kernels as KN
from surreal_amd.model.cnn_stem import CnnParams, CnnStem


class DiagGauss(object):
    """Diagonal Gaussian policy head.  The likelihood / KL / entropy arithmetic of the reference
    class (ppo_net.py:29-72) runs inside the fused loss kernels (csrc/smx_ppo.hip); the
    host-side members used by the rollout workers (ppo_net.py:74-91) are kept here."""

    def __init__(self, action_dim):
        self.d = action_dim

    def sample(self, prob):                      # ppo_net.py:74-83
        prob = np.asarray(prob)
        if prob.ndim == 3:
            prob = prob.reshape(-1, self.d * 2)
        mean_nd, std_nd = prob[:, :self.d], prob[:, self.d:]
        return np.random.randn(prob.shape[0], self.d) * std_nd + mean_nd

    def maxprob(self, prob):                     # ppo_net.py:85-91 (3-D branch quirk kept)
        prob = np.asarray(prob)
        if prob.ndim == 3:
            return prob[:, :, self.d]
        return prob[:, :self.d]


class ZFilter(object):
    """running whitening filter, buffers in HBM (z_filter.py:23-42)"""

    def __init__(self, obs_spec, eps=1e-5, device=None, kernels=None):
        self.K = kernels or KN.default_kernels()
        device = device or KN.default_device()
        self.eps = eps
        self.obs_spec = obs_spec
        self.in_size = sum(obs_spec['low_dim'][k][0] for k in obs_spec['low_dim'].keys())
        d = self.in_size
        self.running_sum = torch.zeros(d, device=device)
        self.running_sumsq = eps * torch.ones(d, device=device)
        self.count = torch.tensor([eps], dtype=torch.float32, device=device)
        self._mean = torch.empty(d, device=device)
        self._std = torch.empty(d, device=device)

    def refresh_stats(self):
        """mean/std used by forward() (z_filter.py:74-76), recomputed on the device"""
        self.K.zfilter_stats(self.running_sum, self.running_sumsq, self.count, self.eps,
                             self._mean, self._std)
        return self._mean, self._std

    def forward(self, inputs):                   # z_filter.py:59-79
        if inputs is None:
            return None
        shape = inputs.shape
        x = inputs.reshape(-1, shape[-1]).contiguous()
        self.refresh_stats()
        out = torch.empty_like(x)
        self.K.zfilter_forward(x, self._mean, self._std, out)
        return out.view(shape)

    def z_update(self, x, count_rows=None):      # z_filter.py:44-57
        """x: [rows, in_size] (a strided 2-D view is fine) or [B, T, in_size]"""
        if x is None:
            return
        if x.dim() == 3:
            x = x.reshape(-1, self.in_size)
        self.K.zfilter_update(x, self.running_sum, self.running_sumsq, self.count,
                              x.shape[0] if count_rows is None else count_rows)

    def running_mean(self):                      # z_filter.py:81-88
        return (self.running_sum / self.count).cpu().numpy()

    def running_std(self):                       # z_filter.py:90-98
        return ((self.running_sumsq / self.count)
                - (self.running_sum / self.count).pow(2)).pow(0.5).cpu().numpy()

    def running_square(self):                    # z_filter.py:100-107
        return (self.running_sumsq / self.count).cpu().numpy()

    def state_dict(self):
        return collections.OrderedDict(running_sum=self.running_sum, running_sumsq=self.running_sumsq,
                                       count=self.count)

    def load_state_dict(self, sd):
        for k in ('running_sum', 'running_sumsq', 'count'):
            src = sd[k]
            if not torch.is_tensor(src):
                src = torch.as_tensor(np.asarray(src), dtype=torch.float32)
            getattr(self, k).copy_(src.reshape(getattr(self, k).shape))


class Mlp3Params(object):
    """views of one three-layer MLP inside a flat parameter buffer + its C descriptor"""

    def __init__(self, flat, offset, D, H1, H2, OUT):
        self.D, self.H1, self.H2, self.OUT = D, H1, H2, OUT
        sizes = [('W1', (H1, D)), ('b1', (H1,)), ('W2', (H2, H1)), ('b2', (H2,)),
                 ('W3', (OUT, H2)), ('b3', (OUT,))]
        self.views = collections.OrderedDict()
        o = offset
        for name, shp in sizes:
            n = int(np.prod(shp))
            self.views[name] = flat[o:o + n].view(*shp)
            o += n
        self.numel = o - offset
        self.offset = offset
        self.desc = L.Mlp3(*(ctypes.c_void_p(self.views[k].data_ptr())
                             for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')), D, H1, H2, OUT)

    @staticmethod
    def count(D, H1, H2, OUT):
        return H1 * D + H1 + H2 * H1 + H2 + OUT * H2 + OUT


class LstmParams(object):
    """views of torch.nn.LSTM's parameters (1 layer) inside a flat buffer + smx_lstm_t"""

    NAMES = ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')

    def __init__(self, flat, offset, D, H):
        self.D, self.H = D, H
        sizes = [('weight_ih', (4 * H, D)), ('weight_hh', (4 * H, H)), ('bias_ih', (4 * H,)),
                 ('bias_hh', (4 * H,))]
        self.views = collections.OrderedDict()
        o = offset
        for name, shp in sizes:
            n = int(np.prod(shp))
            self.views[name] = flat[o:o + n].view(*shp)
            o += n
        self.numel = o - offset
        self.offset = offset
        self.desc = L.Lstm(*(ctypes.c_void_p(self.views[k].data_ptr()) for k in self.NAMES), D, H)

    @staticmethod
    def count(D, H):
        return 4 * H * D + 4 * H * H + 8 * H


class PPOModel(object):
    """
    Actor + critic + z-filter (+ LSTM stem) with the reference constructor signature
    (ppo_net.py:110-118).  All parameters live in ONE flat buffer laid out
    ``[actor MLP | log_var | CNN stem | LSTM layers | critic MLP]`` so that the two optimiser groups
    of the reference -- actor + shared stems, critic + shared stems (ppo_net.py:202-224) -- are the
    contiguous slices ``actor_flat`` and ``critic_flat``.  The CNN stem (pixel observations,
    model/cnn_stem.py) and stacked LSTM layers (rnn_layer > 1) chain the same kernels.
    """

    def __init__(self, obs_spec, action_dim, model_config, use_cuda=True, init_log_sig=0,
                 use_z_filter=False, if_pixel_input=False, rnn_config=None, device=None,
                 kernels=None):
        self.K = kernels or KN.default_kernels()     # raises without the HIP library / a GPU
        device = device or KN.default_device()
        self.if_pixel = bool(if_pixel_input)
        self.if_rnn = bool(rnn_config is not None and rnn_config.get('if_rnn_policy', False))
        self.rnn_layers = int(rnn_config.get('rnn_layer', 1)) if self.if_rnn else 0
        if self.if_rnn and self.rnn_layers < 1:
            raise ValueError('rnn_layer must be >= 1')
        self.obs_spec = obs_spec
        self.action_dim = action_dim
        self.model_config = model_config
        self.use_z_filter = use_z_filter
        self.init_log_sig = init_log_sig
        self.if_pixel_input = if_pixel_input
        self.rnn_config = rnn_config
        self.device = device
        self.low_dim = 0
        if 'low_dim' in obs_spec.keys():
            for key in obs_spec['low_dim'].keys():
                self.low_dim += obs_spec['low_dim'][key][0]
        D, A = self.low_dim, action_dim
        ah, ch = model_config['actor_fc_hidden_sizes'], model_config['critic_fc_hidden_sizes']
        # The LSTM kernels read W_hh rows and the hidden state 16 bytes at a time: a hidden size that is not a multiple of 4
        # (ppo_net.py:144-149 takes any) is PADDED inside the parameter layout -- every gate block, the recurrent columns,
        # the MLPs' input columns -- with zeros.  A padded unit's gates see zero pre-activations (g = tanh(0) = 0, so its
        # cell stays at the zero it starts from and its output is 0), nothing downstream reads it through a non-zero
        # weight, so its gradients are exactly zero and Adam leaves the zeros alone.  `rnn_hidden` is the PADDED size (what
        # every device buffer is shaped by); `rnn_hidden_logical` what the configuration, the parameter dict, the wire
        # formats and the agents' cells see.
        self.rnn_hidden_logical = int(rnn_config.rnn_hidden) if self.if_rnn else 0
        self.rnn_hidden = (self.rnn_hidden_logical + 3) & ~3
        self.cnn_feature_dim = int(model_config['cnn_feature_dim']) if self.if_pixel else 0
        Dx = D + self.cnn_feature_dim                    # stem input width (ppo_net.py:145)
        self.stem_in = Dx
        F = self.rnn_hidden if self.if_rnn else Dx       # input width of the MLPs (ppo_net.py:155-156)
        # the actor block is padded to a 16-byte boundary (the pad has zero gradient, so Adam
        # leaves it alone): the GEMM kernels take 16-byte operand loads only from aligned bases
        n_actor = (Mlp3Params.count(F, ah[0], ah[1], A) + A + 3) & ~3
        cam = tuple(obs_spec['pixel']['camera0']) if self.if_pixel else None
        n_cnn = CnnParams.count(cam, self.cnn_feature_dim) if self.if_pixel else 0
        # stacked LSTM layers (nn.LSTM(in, hid, rnn_layer), ppo_net.py:146): layer l > 0 reads layer l-1's output
        rnn_in = [Dx] + [self.rnn_hidden] * max(0, self.rnn_layers - 1)
        rnn_n = [LstmParams.count(d, self.rnn_hidden) for d in rnn_in] if self.if_rnn else []
        n_rnn = sum(rnn_n)
        n_critic = Mlp3Params.count(F, ch[0], ch[1], 1)
        self.flat = torch.zeros(n_actor + n_cnn + n_rnn + n_critic, device=device)
        self.n_actor_block, self.n_cnn, self.n_rnn, self.n_stem = n_actor, n_cnn, n_rnn, n_cnn + n_rnn
        self.actor_flat = self.flat[:n_actor + self.n_stem]   # optimiser group: actor (+ shared stems)
        self.critic_flat = self.flat[n_actor:]                # optimiser group: (shared stems +) critic
        self.actor = Mlp3Params(self.flat, 0, F, ah[0], ah[1], A)
        self.cnn = CnnParams(self.flat, n_actor, cam, self.cnn_feature_dim) if self.if_pixel else None
        self.rnns, o = [], n_actor + n_cnn
        for d, cnt in zip(rnn_in if self.if_rnn else [], rnn_n):
            self.rnns.append(LstmParams(self.flat, o, d, self.rnn_hidden))
            o += cnt
        self.rnn = self.rnns[0] if self.if_rnn else None        # (the first layer, what sits on the stem input)
        self.rnn_offsets = np.cumsum([0] + rnn_n[:-1]).tolist() if self.if_rnn else []   # within the LSTM block
        self.rnn_counts = rnn_n
        self.critic = Mlp3Params(self.flat, n_actor + self.n_stem, F, ch[0], ch[1], 1)
        self._cnn_stem = CnnStem(self.K) if self.if_pixel else None
        self.log_var = self.flat[self.actor.numel:self.actor.numel + A].view(1, A)
        self._init_parameters()
        if use_z_filter:
            assert self.low_dim > 0, 'No low dimensional input, please turn off z-filter'
            self.z_filter = ZFilter(obs_spec, device=device, kernels=self.K)

    def _init_parameters(self):
        # torch.nn.Linear default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torchx's own init is
        # unknown -- source absent; parity tests always inject parameters)
        for net in (self.actor, self.critic):
            for name, v in net.views.items():
                fan_in = net.views['W' + name[1]].shape[1]
                v.uniform_(-1.0 / np.sqrt(fan_in), 1.0 / np.sqrt(fan_in))
        self.log_var.fill_(float(self.init_log_sig))
        if self.if_pixel:
            self.cnn.init_torch_default()
        if self.if_rnn:                          # torch.nn.LSTM default: U(-1/sqrt(H), 1/sqrt(H))
            b = 1.0 / np.sqrt(self.rnn_hidden_logical)
            for k, v in self.named_parameters().items():        # (the logical views: a pad stays zero)
                if k.startswith('rnn.'):
                    v.uniform_(-b, b)
        if self.if_rnn and self.rnn_hidden != self.rnn_hidden_logical:
            for net in (self.actor, self.critic):               # the MLPs' input columns behind the logical width
                net.views['W1'][:, self.rnn_hidden_logical:].zero_()

    # ---- canonical parameter dict (names shared with the oracle / synthetic generator) ----
    def named_parameters(self):
        out = collections.OrderedDict()
        for net, pre in ((self.actor, 'actor'), (self.critic, 'critic')):
            for i in (1, 2, 3):
                out['%s.fc%d.W' % (pre, i)] = net.views['W%d' % i]
                out['%s.fc%d.b' % (pre, i)] = net.views['b%d' % i]
            if pre == 'actor':
                out['actor.log_var'] = self.log_var
        if self.if_pixel:
            for k, v in self.cnn.views.items():
                out['cnn.' + k] = v
        H, Hp = self.rnn_hidden_logical, self.rnn_hidden
        padded = self.if_rnn and H != Hp
        if padded:                                           # the MLPs read the LSTM output: logical columns only
            out['actor.fc1.W'] = self.actor.views['W1'][:, :H]
            out['critic.fc1.W'] = self.critic.views['W1'][:, :H]
        if self.if_rnn:
            for layer, r in enumerate(self.rnns):            # nn.LSTM's names: *_l0 (kept bare), *_l1, ...
                for k, v in r.views.items():
                    if padded:                               # [4 gate blocks of H of Hp rows (, H of Hp recurrent columns)]
                        v = v.view((4, Hp) + tuple(v.shape[1:]))[:, :H]
                        if k == 'weight_hh' or (k == 'weight_ih' and layer > 0):
                            v = v[:, :, :H]
                    out['rnn.' + k + ('' if layer == 0 else '_l%d' % layer)] = v
        return out

    def _logical_shape(self, k, v):
        """the shape a parameter has outside this class (torch.nn.LSTM's [4H, .]): differs from the view's only under padding"""
        if k.startswith('rnn.') and v.dim() >= 2 and v.shape[0] == 4 and self.rnn_hidden != self.rnn_hidden_logical:
            return (4 * v.shape[1],) + tuple(v.shape[2:])
        return tuple(v.shape)

    def load_params(self, params):
        with torch.no_grad():
            for k, v in self.named_parameters().items():
                v.copy_(torch.as_tensor(np.asarray(params[k]), dtype=torch.float32).view(v.shape))

    def numpy_params(self):
        return collections.OrderedDict((k, v.detach().cpu().numpy().reshape(self._logical_shape(k, v)).copy())
                                       for k, v in self.named_parameters().items())

    def state_dict(self):
        # (under padding the LSTM entries are reshaped COPIES in torch.nn.LSTM's shapes: an export, not an alias)
        sd = collections.OrderedDict((k, v if tuple(v.shape) == self._logical_shape(k, v) else v.reshape(self._logical_shape(k, v)))
                                     for k, v in self.named_parameters().items())
        if self.use_z_filter:
            for k, v in self.z_filter.state_dict().items():
                sd['z_filter.' + k] = v
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, v in self.named_parameters().items():
                src = sd[k]
                if torch.is_tensor(src):          # device -> device: the in-process "PS"
                    v.copy_(src.view(v.shape))
                else:
                    v.copy_(torch.as_tensor(np.asarray(src), dtype=torch.float32).view(v.shape))
        if self.use_z_filter:
            self.z_filter.load_state_dict({k[len('z_filter.'):]: v for k, v in sd.items()
                                           if k.startswith('z_filter.')})

    def get_actor_params(self):                  # ppo_net.py:202-212
        return [self.actor_flat]

    def get_critic_params(self):                 # ppo_net.py:214-224
        return [self.critic_flat]

    def update_target_params(self, net):         # ppo_net.py:226-242
        self.flat.copy_(net.flat)
        if self.use_z_filter:
            self.z_filter.load_state_dict(net.z_filter.state_dict())

    def update_target_z_filter(self, net):       # ppo_net.py:244-251
        if self.use_z_filter:
            self.z_filter.load_state_dict(net.z_filter.state_dict())

    def _gather_low_dim_input(self, obs):        # ppo_net.py:168-178
        if 'low_dim' not in obs.keys():
            return None
        parts = [obs['low_dim'][k] for k in obs['low_dim'].keys()]
        return parts[0] if len(parts) == 1 else torch.cat(parts, -1)

    def _mlp(self, net, x, out_act):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        rows = x2.shape[0]
        h1 = torch.empty(rows, net.H1, device=x.device)
        h2 = torch.empty(rows, net.H2, device=x.device)
        out = torch.empty(rows, net.OUT, device=x.device)
        self.K.mlp3_forward(net, x2, h1, h2, out, out_act)
        return out, shape

    def _stem(self, obs, cells, want_cells=False):
        """low-dim concat -> z-filter -> [LSTM]  (ppo_net.py:262-279).  With the LSTM stem the
        input is (B, T, D) and cells = (h0, c0), each (1, B, H), or None for zeros."""
        x = self._gather_low_dim_input(obs)
        if self.use_z_filter:
            x = self.z_filter.forward(x)
        if self.if_pixel:                        # ppo_net.py:268-275: cat(z(low_dim), cnn(camera0 / 255))
            pix = obs['pixel']['camera0']
            lead = tuple(pix.shape[:-3])
            frames = pix.reshape((-1,) + tuple(pix.shape[-3:])).contiguous()
            nF = frames.shape[0]
            xin = torch.empty(nF, self.stem_in, device=frames.device)
            xin[:, :self.low_dim].copy_(x.reshape(nF, self.low_dim))
            cws = CnnStem.workspace(self.cnn, nF, frames.device, backward=False)
            self._cnn_stem.forward(self.cnn, frames, nF, cws, xin[:, self.low_dim:])
            x = xin.view(lead + (self.stem_in,))
        if not self.if_rnn:
            return x, cells
        assert x.dim() == 3, 'the LSTM stem takes (batch, time, features) observations'
        B, T, D = x.shape
        H = self.rnn_hidden
        dev = x.device
        x2 = x.reshape(B * T, D).contiguous()
        nl = self.rnn_layers
        h0 = c0 = None
        Hl = self.rnn_hidden_logical
        if cells is not None:                    # (layers, B, H) each, H the configured size (zero padded for the kernels)
            h0, c0 = torch.zeros(nl, B, H, device=dev), torch.zeros(nl, B, H, device=dev)
            h0[:, :, :Hl].copy_(cells[0].reshape(nl, B, Hl))
            c0[:, :, :Hl].copy_(cells[1].reshape(nl, B, Hl))
        gates = torch.empty(B * T, 4 * H, device=dev)
        cs = torch.empty(B * T, H, device=dev)
        hN = cN = None
        if want_cells:
            hN, cN = torch.empty(nl, B, H, device=dev), torch.empty(nl, B, H, device=dev)
        for layer, rnn in enumerate(self.rnns):  # layer l reads layer l-1's output sequence
            out = torch.empty(B, T, H, device=dev)
            self.K.lstm_forward(rnn, x2, B, T, None if h0 is None else h0[layer],
                                None if c0 is None else c0[layer], gates, out, cs, None,
                                hN[layer] if want_cells else None, cN[layer] if want_cells else None)
            x2 = out.view(B * T, H)
        if want_cells:
            cells = (hN[:, :, :Hl], cN[:, :, :Hl]) if Hl != H else (hN, cN)
        return out, cells

    def forward_actor(self, obs, cells=None):    # ppo_net.py:253-282, builders.py:114-132
        x, _ = self._stem(obs, cells)
        mean, shape = self._mlp(self.actor, x, L.SMX_ACT_TANH)
        std = torch.exp(self.log_var) * torch.ones_like(mean)
        action = torch.cat((mean, std), dim=1)
        if len(shape) == 3:
            action = action.view(shape[0], shape[1], -1)
        return action

    def forward_critic(self, obs, cells=None):   # ppo_net.py:284-315, builders.py:159-175
        x, _ = self._stem(obs, cells)
        v, shape = self._mlp(self.critic, x, L.SMX_ACT_NONE)
        if len(shape) == 3:
            v = v.view(shape[0], shape[1], 1)
        return v

    def forward_actor_expose_cells(self, obs, cells=None):   # ppo_net.py:317-354
        """one environment step for n actors: obs (n, D), cells (1, n, H) x 2 -> (pd (n, 2A),
        new cells).  The reference is the n = 1 case (`obs.view(1, 1, -1)`, :338)."""
        if not self.if_rnn:
            return self.forward_actor(obs, cells), cells
        step = {mod: {k: v.unsqueeze(1) for k, v in obs[mod].items()} for mod in obs.keys()}
        x, cells = self._stem(step, cells, want_cells=True)
        mean, _ = self._mlp(self.actor, x.reshape(-1, self.rnn_hidden), L.SMX_ACT_TANH)
        std = torch.exp(self.log_var) * torch.ones_like(mean)
        return torch.cat((mean, std), dim=1), cells

    def z_update(self, obs):                     # ppo_net.py:356-366
        if not self.use_z_filter:
            raise ValueError('Z_update called when network is set to not use z_filter')
        self.z_filter.z_update(self._gather_low_dim_input(obs))
