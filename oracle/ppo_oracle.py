"""
TEST INFRASTRUCTURE ONLY.  CPU restatement (PyTorch fp32, CPU tensors) of the
reference's PPO learner hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product path (surreal_amd/)
never does and fails loudly when its HIP library is missing.

Every function cites the reference file:line it restates (paths relative to
/root/reference).  The arithmetic is issued through the *same* ATen ops the
reference calls (it is pure Python on torch), so on identical inputs and
identical injected parameters this file reproduces the reference bit-for-bit
on CPU; oracle/gen_golden.py asserts that against the reference's own code run
under oracle/ref_shims.py and commits the vectors under tests/golden/.

Parity status: the reference's own tests pin nothing at this boundary
("parity unpinned" by reference tests); the pin is the
reference source executed here -> tests/golden/*.npz.
"""
import collections
import itertools

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# DiagGauss  (surreal/model/ppo_net.py:13-91)
# --------------------------------------------------------------------------
class DiagGauss(object):
    def __init__(self, action_dim):
        self.d = action_dim

    def loglikelihood(self, a, prob):            # ppo_net.py:29-40
        if a.dim() == 3:
            a = a.reshape(-1, self.d)
            prob = prob.reshape(-1, 2 * self.d)
        mean0 = prob[:, :self.d]
        std0 = prob[:, self.d:]
        return (-0.5 * (((a - mean0) / std0).pow(2)).sum(dim=1, keepdim=True)
                - 0.5 * np.log(2.0 * np.pi) * self.d
                - std0.log().sum(dim=1, keepdim=True))

    def likelihood(self, a, prob):               # ppo_net.py:42-46
        return torch.clamp(self.loglikelihood(a, prob).exp(), min=1e-5)

    def kl(self, prob0, prob1):                  # ppo_net.py:48-62
        if prob0.dim() == 3:
            prob0 = prob0.reshape(-1, 2 * self.d)
            prob1 = prob1.reshape(-1, 2 * self.d)
        mean0, std0 = prob0[:, :self.d], prob0[:, self.d:]
        mean1, std1 = prob1[:, :self.d], prob1[:, self.d:]
        return ((std1 / std0).log()).sum(dim=1) + (
            (std0.pow(2) + (mean0 - mean1).pow(2)) / (2.0 * std1.pow(2))).sum(dim=1) \
            - 0.5 * self.d

    def entropy(self, prob):                     # ppo_net.py:64-72 (formula sic)
        if prob.dim() == 3:
            prob = prob.reshape(-1, 2 * self.d)
        std_nd = prob[:, self.d:]
        return 0.5 * std_nd.log().sum(dim=1) + .5 * np.log(2 * np.pi * np.e) * self.d


# --------------------------------------------------------------------------
# ZFilter  (surreal/model/z_filter.py:7-107)
# --------------------------------------------------------------------------
class ZFilter(object):
    def __init__(self, in_size, eps=1e-5, state=None):
        self.eps = eps
        self.in_size = in_size
        # z_filter.py:40-42
        self.running_sum = torch.zeros(in_size)
        self.running_sumsq = eps * torch.ones(in_size)
        self.count = torch.tensor([eps], dtype=torch.get_default_dtype())
        if state is not None:
            self.load(state)

    def load(self, state):
        self.running_sum = torch.tensor(np.asarray(state['running_sum']), dtype=torch.get_default_dtype()).clone()
        self.running_sumsq = torch.tensor(np.asarray(state['running_sumsq']), dtype=torch.get_default_dtype()).clone()
        self.count = torch.tensor(np.asarray(state['count']), dtype=torch.get_default_dtype()).clone()

    def state(self):
        return {'running_sum': self.running_sum.numpy().copy(),
                'running_sumsq': self.running_sumsq.numpy().copy(),
                'count': self.count.numpy().copy()}

    def z_update(self, x):                       # z_filter.py:44-57
        if x.dim() == 3:
            x = x.reshape(-1, self.in_size)
        self.running_sum += torch.sum(x, dim=0)
        self.running_sumsq += torch.sum(x * x, dim=0)
        self.count += float(len(x))

    def forward(self, inputs):                   # z_filter.py:59-79
        shape = inputs.size()
        inputs = inputs.reshape(-1, shape[-1])
        running_mean = self.running_sum / self.count
        running_std = torch.clamp(
            (self.running_sumsq / self.count - running_mean.pow(2)).pow(0.5), min=self.eps)
        normed = torch.clamp((inputs - running_mean) / running_std, -5.0, 5.0)
        return normed.view(shape)

    def running_mean(self):                      # z_filter.py:81-88
        return (self.running_sum / self.count).numpy()

    def running_std(self):                       # z_filter.py:90-98
        return ((self.running_sumsq / self.count)
                - (self.running_sum / self.count).pow(2)).pow(0.5).numpy()

    def running_square(self):                    # z_filter.py:100-107
        return (self.running_sumsq / self.count).numpy()


# --------------------------------------------------------------------------
# RewardFilter  (surreal/model/reward_filter.py:5-63), quirk at :42 kept
# --------------------------------------------------------------------------
class RewardFilter(object):
    def __init__(self, eps=1e-5):
        self.eps = eps
        self.count = torch.tensor(eps, dtype=torch.get_default_dtype())
        self.running_sum = torch.tensor(0.0, dtype=torch.get_default_dtype())
        self.running_sumsq = torch.tensor(0.0, dtype=torch.get_default_dtype())

    def update(self, x):                         # reward_filter.py:33-42
        self.count += float(np.prod(x.size()))
        self.running_sum += x.sum()
        self.running_sumsq = (x * x).sum()       # overwrite (sic)

    def forward(self, inputs):                   # reward_filter.py:44-57
        mean = self.running_sum / self.count
        std = torch.clamp((self.running_sumsq / self.count - mean.pow(2)).pow(0.5),
                          min=self.eps)
        return torch.clamp((inputs - mean) / std, -5.0, 5.0)

    def reward_mean(self):                       # reward_filter.py:59-63
        return (self.running_sum / self.count).item()


# --------------------------------------------------------------------------
# PPOModel  (surreal/model/ppo_net.py:94-375; builders.py:86-175)
# --------------------------------------------------------------------------
class OraclePPOModel(object):
    """
    Parameters are the canonical flat dict of surreal_amd.synthetic.make_ppo_params:
    actor.fc{1,2,3}.{W,b}, actor.log_var, critic.fc{1,2,3}.{W,b}, optional
    rnn.{weight_ih,weight_hh,bias_ih,bias_hh} (single-layer LSTM, batch_first;
    ppo_net.py:143-152).
    """

    def __init__(self, params, action_dim, use_z_filter=True, zstate=None, in_size=None):
        self.p = collections.OrderedDict(
            (k, torch.tensor(np.asarray(v), dtype=torch.get_default_dtype()).clone().requires_grad_(True))
            for k, v in params.items())
        self.action_dim = action_dim
        self.use_z_filter = use_z_filter
        self.if_rnn = 'rnn.weight_ih' in self.p
        if self.if_rnn:
            hid, din = self.p['rnn.weight_ih'].shape[0] // 4, self.p['rnn.weight_ih'].shape[1]
            layers = 1 + sum(1 for k in self.p if k.startswith('rnn.weight_ih_l'))   # rnn_layer (ppo_net.py:146)
            self._rnn = nn.LSTM(din, hid, layers, batch_first=True)
            for layer in range(layers):
                sfx = '' if layer == 0 else '_l%d' % layer
                for nm in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                    mod_p = getattr(self._rnn, '%s_l%d' % (nm, layer))
                    with torch.no_grad():
                        mod_p.copy_(self.p['rnn.' + nm + sfx])
                    self.p['rnn.' + nm + sfx] = mod_p    # the canonical dict aliases the module's Parameters
        # optional CNN stem (builders.py:8-33): Conv2d(16,k8,s4)-ReLU-Conv2d(32,k4,s2)-ReLU-Flatten-
        # Linear(cnn_feature_dim)-ReLU; the same ATen conv2d / linear ops the reference calls
        self.if_pixel = 'cnn.conv1.W' in self.p
        feat = 0
        if self.if_pixel:
            w1, w2, wf = self.p['cnn.conv1.W'], self.p['cnn.conv2.W'], self.p['cnn.fc.W']
            feat = wf.shape[0]
            self._cnn = nn.Sequential(nn.Conv2d(w1.shape[1], w1.shape[0], 8, 4), nn.ReLU(),
                                      nn.Conv2d(w2.shape[1], w2.shape[0], 4, 2), nn.ReLU(),
                                      nn.Flatten(), nn.Linear(wf.shape[1], feat), nn.ReLU())
            mods = {'conv1': self._cnn[0], 'conv2': self._cnn[2], 'fc': self._cnn[5]}
            with torch.no_grad():
                for nm, mod in mods.items():
                    mod.weight.copy_(self.p['cnn.%s.W' % nm])
                    mod.bias.copy_(self.p['cnn.%s.b' % nm])
            for nm, mod in mods.items():
                self.p['cnn.%s.W' % nm] = mod.weight
                self.p['cnn.%s.b' % nm] = mod.bias
        if use_z_filter:
            d = in_size if in_size is not None else (
                self.p['rnn.weight_ih'].shape[1] if self.if_rnn else self.p['actor.fc1.W'].shape[1]) - feat
            self.z_filter = ZFilter(d, state=zstate)

    # ppo_net.py:202-224 -- shared stems belong to BOTH parameter groups
    def actor_params(self):
        # nn.Module.parameters() yields the module's own parameters (log_var,
        # builders.py:112) before its children's (self.model.*)
        names = ['actor.log_var'] + [k for k in self.p
                                     if k.startswith('actor.') and k != 'actor.log_var']
        names += self._cnn_names()
        if self.if_rnn:
            names += [k for k in self.p if k.startswith('rnn.')]
        return [self.p[k] for k in names]

    def _cnn_names(self):
        # module order of cnn_stem.parameters(): conv1, conv2, fc (weight then bias each)
        if not self.if_pixel:
            return []
        return ['cnn.%s.%s' % (m, q) for m in ('conv1', 'conv2', 'fc') for q in ('W', 'b')]

    def critic_params(self):
        names = [k for k in self.p if k.startswith('critic.')]
        names += self._cnn_names()
        if self.if_rnn:
            names += [k for k in self.p if k.startswith('rnn.')]
        return [self.p[k] for k in names]

    def clear_actor_grad(self):                  # ppo_net.py:180-189
        for q in self.actor_params():
            q.grad = None

    def clear_critic_grad(self):                 # ppo_net.py:191-200
        for q in self.critic_params():
            q.grad = None

    def numpy_params(self):
        return collections.OrderedDict((k, v.detach().numpy().copy()) for k, v in self.p.items())

    def load_from(self, other):                  # ppo_net.py:226-242
        with torch.no_grad():
            for k in self.p:
                self.p[k].copy_(other.p[k])
        if self.use_z_filter:
            self.z_filter.load(other.z_filter.state())

    def _stem(self, obs, cells):
        # ppo_net.py:262-279: concat low-dim keys -> z-filter -> [LSTM]
        x = torch.cat([obs['low_dim'][k] for k in obs['low_dim'].keys()], -1)
        if self.use_z_filter:
            x = self.z_filter.forward(x)
        if self.if_pixel:                        # ppo_net.py:268-275, 368-375; builders.py:23-33
            pix = obs['pixel']['camera0'] / 255.0
            shp = pix.size()
            if len(shp) == 5:
                pix = pix.view(-1, *shp[2:])
            f = self._cnn(pix)
            if len(shp) == 5:
                f = f.view(shp[0], shp[1], -1)
            x = torch.cat([x, f], dim=-1)
        if self.if_rnn:
            x = self._lstm(x, cells)
        return x

    def _lstm(self, x, cells):
        """ppo_net.py:146-149,277-279: nn.LSTM(in, hid, 1, batch_first=True) -- the same
        ATen fused-LSTM op the reference calls (a hand-rolled cell loop differs by 1 ulp)"""
        out, _ = self._rnn(x, cells)
        return out.contiguous()

    def forward_actor(self, obs, cells=None):    # ppo_net.py:253-282, builders.py:114-132
        x = self._stem(obs, cells)
        shape = x.size()
        hi = (x.dim() == 3)
        if hi:
            x = x.reshape(-1, shape[2])
        p = self.p
        h = torch.relu(F.linear(x, p['actor.fc1.W'], p['actor.fc1.b']))
        h = torch.relu(F.linear(h, p['actor.fc2.W'], p['actor.fc2.b']))
        mean = torch.tanh(F.linear(h, p['actor.fc3.W'], p['actor.fc3.b']))
        std = torch.exp(p['actor.log_var']) * torch.ones(mean.size())
        action = torch.cat((mean, std), dim=1)
        if hi:
            action = action.view(shape[0], shape[1], -1)
        return action

    def forward_critic(self, obs, cells=None):   # ppo_net.py:284-315, builders.py:159-175
        x = self._stem(obs, cells)
        shape = x.size()
        hi = (x.dim() == 3)
        if hi:
            x = x.reshape(-1, shape[2])
        p = self.p
        h = torch.relu(F.linear(x, p['critic.fc1.W'], p['critic.fc1.b']))
        h = torch.relu(F.linear(h, p['critic.fc2.W'], p['critic.fc2.b']))
        v = F.linear(h, p['critic.fc3.W'], p['critic.fc3.b'])
        if hi:
            v = v.view(shape[0], shape[1], 1)
        return v

    def z_update(self, obs):                     # ppo_net.py:356-366
        x = torch.cat([obs['low_dim'][k] for k in obs['low_dim'].keys()], -1)
        self.z_filter.z_update(x)


DEFAULT_HYPER = dict(   # surreal/main/ppo_configs.py:15-94
    gamma=0.995, lam=0.97, n_step=25, horizon=5, if_rnn_policy=False,
    norm_adv=True, use_z_filter=True, use_r_filter=False, reward_scale=1.0,
    ppo_mode='adapt', lr_actor=1e-4, lr_critic=1e-4,
    clip_actor_gradient=True, actor_gradient_norm_clip=5.0,
    clip_critic_gradient=True, critic_gradient_norm_clip=5.0,
    actor_regularization=0.0, critic_regularization=0.0,
    epoch_policy=10, epoch_baseline=10, kl_target=0.015,
    adjust_threshold=(0.5, 2.0),
    kl_cutoff_coeff=250.0, beta_init=1.0, beta_range=(1.0 / 35.0, 35.0), adapt_scale=1.5,
    clip_epsilon_init=0.2, clip_range=(0.05, 0.3), clip_scale=1.2,
)


class OraclePPOLearner(object):
    """restates PPOLearner numerics (surreal/learner/ppo.py:194-666)"""

    def __init__(self, params, action_dim, batch_size, zstate=None, **hyper):
        h = dict(DEFAULT_HYPER)
        h.update(hyper)
        self.h = h
        for k, v in h.items():
            setattr(self, k, v)
        self.batch_size = batch_size
        self.action_dim = action_dim
        self.model = OraclePPOModel(params, action_dim, self.use_z_filter, zstate)
        self.ref_target_model = OraclePPOModel(params, action_dim, self.use_z_filter, zstate)
        self.ref_target_model.load_from(self.model)            # ppo.py:151
        # ppo.py:159-168
        self.critic_optim = torch.optim.Adam(self.model.critic_params(), lr=self.lr_critic,
                                             weight_decay=self.critic_regularization)
        self.actor_optim = torch.optim.Adam(self.model.actor_params(), lr=self.lr_actor,
                                            weight_decay=self.actor_regularization)
        self.pd = DiagGauss(action_dim)
        self.cells = None
        if self.ppo_mode == 'adapt':                           # ppo.py:108-118
            self.beta = self.beta_init
            self.eta = self.kl_cutoff_coeff
        else:
            self.clip_epsilon = self.clip_epsilon_init
        if self.use_r_filter:
            self.reward_filter = RewardFilter()
        self.kl_record = []
        self.exp_counter = 0
        self.trace = None

    # ---------------------------------------------------------------- losses
    def _clip_loss(self, obs, actions, advantages, behave_pol):   # ppo.py:194-225
        learn_pol = self.model.forward_actor(obs, self.cells)
        learn_prob = self.pd.likelihood(actions, learn_pol)
        behave_prob = self.pd.likelihood(actions, behave_pol)
        prob_ratio = learn_prob / behave_prob
        cliped_ratio = torch.clamp(prob_ratio, 1 - self.clip_epsilon, 1 + self.clip_epsilon)
        surr = -prob_ratio * advantages.view(-1, 1)
        cliped_surr = -cliped_ratio * advantages.view(-1, 1)
        clip_loss = torch.cat([surr, cliped_surr], 1).max(1)[0].mean()
        stats = {
            '_surr_loss': surr.mean().item(),
            '_clip_surr_loss': clip_loss.item(),
            '_entropy': self.pd.entropy(learn_pol).mean().item(),
            '_clip_epsilon': self.clip_epsilon,
        }
        return clip_loss, stats

    def _adapt_loss(self, obs, actions, advantages, behave_pol, ref_pol):  # ppo.py:250-285
        learn_pol = self.model.forward_actor(obs, self.cells)
        prob_behave = self.pd.likelihood(actions, behave_pol)
        prob_learn = self.pd.likelihood(actions, learn_pol)
        kl = self.pd.kl(ref_pol, learn_pol).mean()
        surr = -(advantages.view(-1, 1) * (prob_learn / torch.clamp(prob_behave, min=1e-2))).mean()
        loss = surr + self.beta * kl
        entropy = self.pd.entropy(learn_pol).mean()
        if kl.item() - 2.0 * self.kl_target > 0:
            loss = loss + self.eta * (kl - 2.0 * self.kl_target).pow(2)
        stats = {
            '_kl_loss_adapt': loss.item(),
            '_surr_loss': surr.item(),
            '_pol_kl': kl.item(),
            '_entropy': entropy.item(),
            '_beta': self.beta,
        }
        return loss, stats

    def _policy_update(self, obs, actions, advantages, behave_pol, ref_pol):
        # ppo.py:227-248 / 287-309
        if self.ppo_mode == 'clip':
            loss, stats = self._clip_loss(obs, actions, advantages, behave_pol)
        else:
            loss, stats = self._adapt_loss(obs, actions, advantages, behave_pol, ref_pol)
        self.model.clear_actor_grad()
        loss.backward()
        if self.clip_actor_gradient:
            stats['grad_norm_actor'] = float(nn.utils.clip_grad_norm_(
                self.model.actor_params(), self.actor_gradient_norm_clip))
        self.actor_optim.step()
        return stats

    def _value_loss(self, obs, returns):                         # ppo.py:311-332
        values = self.model.forward_critic(obs, self.cells)
        if values.dim() == 3:
            values = values.squeeze(2)
        explained_var = 1 - torch.var(returns - values) / torch.var(returns)
        loss = (values - returns).pow(2).mean()
        return loss, {'_val_loss': loss.item(), '_val_explained_var': explained_var.item()}

    def _value_update(self, obs, returns):                       # ppo.py:334-353
        loss, stats = self._value_loss(obs, returns)
        self.model.clear_critic_grad()
        loss.backward()
        if self.clip_critic_gradient:
            stats['grad_norm_critic'] = float(nn.utils.clip_grad_norm_(
                self.model.critic_params(), self.critic_gradient_norm_clip))
        self.critic_optim.step()
        return stats

    # ------------------------------------------------------------- GAE
    def _gae_and_return(self, obs, obs_next, rewards, dones):    # ppo.py:355-418
        index_set = torch.tensor(range(self.n_step), dtype=torch.get_default_dtype())
        gamma = torch.pow(self.gamma, index_set)
        lam = torch.pow(self.lam, index_set)
        oc = {}
        for mod in obs.keys():
            oc[mod] = {}
            for k in obs[mod].keys():
                oc[mod][k] = torch.cat([obs[mod][k], obs_next[mod][k]], dim=1)
                if not self.if_rnn_policy:
                    s = oc[mod][k].size()
                    oc[mod][k] = oc[mod][k].view(-1, *s[2:])
        values = self.model.forward_critic(oc, self.cells)
        values = values.view(self.batch_size, self.n_step + 1)
        values = values.detach().clone()
        self.last_values_raw = values.clone()
        values[:, 1:] *= 1 - dones
        return self.gae_from_values(values, rewards, gamma, lam)

    def gae_from_values(self, values, rewards, gamma=None, lam=None):
        """ppo.py:389-418 given the already-masked values (B, N+1)"""
        if gamma is None:
            index_set = torch.tensor(range(self.n_step), dtype=torch.get_default_dtype())
            gamma = torch.pow(self.gamma, index_set)
            lam = torch.pow(self.lam, index_set)
        if self.if_rnn_policy:
            tds = rewards + self.gamma * values[:, 1:] - values[:, :-1]
            eff_len = self.n_step - self.horizon + 1
            gamma = gamma[:self.horizon]
            lam = lam[:self.horizon]
            returns = torch.zeros(self.batch_size, eff_len)
            advs = torch.zeros(self.batch_size, eff_len)
            for step in range(eff_len):
                returns[:, step] = torch.sum(gamma * rewards[:, step:step + self.horizon], 1) + \
                    values[:, step + self.horizon] * (self.gamma ** self.horizon)
                advs[:, step] = torch.sum(tds[:, step:step + self.horizon] * gamma * lam, 1)
            self.last_adv_raw = advs.clone()
            if self.norm_adv:
                std = advs.std()
                mean = advs.mean()
                advs = (advs - mean) / max(std, 1e-4)
            return advs, returns
        returns = torch.sum(gamma * rewards, 1) + values[:, -1] * (self.gamma ** self.n_step)
        tds = rewards + self.gamma * values[:, 1:] - values[:, :-1]
        gae = torch.sum(tds * gamma * lam, 1)
        self.last_adv_raw = gae.clone()
        if self.norm_adv:
            std = gae.std()
            mean = gae.mean()
            gae = (gae - mean) / max(std, 1e-4)
        return gae.view(-1, 1), returns.view(-1, 1)

    # ------------------------------------------------------------- batch
    def _preprocess_batch_ppo(self, batch):                      # ppo.py:420-484
        out = dict(batch)
        obs, obs_next = {}, {}
        for m in batch['obs']:
            obs[m], obs_next[m] = {}, {}
            for k in batch['obs'][m]:
                obs[m][k] = torch.as_tensor(np.asarray(batch['obs'][m][k]), dtype=torch.get_default_dtype()).clone()
                obs_next[m][k] = torch.as_tensor(np.asarray(batch['obs_next'][m][k]), dtype=torch.get_default_dtype()).clone()
        out['obs'], out['obs_next'] = obs, obs_next
        out['actions'] = torch.as_tensor(np.asarray(batch['actions']), dtype=torch.get_default_dtype()).clone()
        rewards = torch.as_tensor(np.asarray(batch['rewards']), dtype=torch.get_default_dtype()) * self.reward_scale
        if self.use_r_filter:
            normed = self.reward_filter.forward(rewards)
            self.reward_filter.update(rewards)
            rewards = normed
        out['rewards'] = rewards
        out['dones'] = torch.as_tensor(np.asarray(batch['dones']), dtype=torch.get_default_dtype()).clone()
        if batch.get('persistent_infos') is not None:
            out['persistent_infos'] = [torch.as_tensor(np.asarray(x), dtype=torch.get_default_dtype()).clone()
                                       for x in batch['persistent_infos']]
        if batch.get('onetime_infos') is not None:
            out['onetime_infos'] = [torch.as_tensor(np.asarray(x), dtype=torch.get_default_dtype()).clone()
                                    for x in batch['onetime_infos']]
        return out

    def _optimize(self, obs, actions, rewards, obs_next, persistent_infos, onetime_infos,
                  dones):                                        # ppo.py:487-586
        trace = {'policy': [], 'value': []}
        pds = persistent_infos[-1]
        if self.if_rnn_policy:
            h = onetime_infos[0].transpose(0, 1).contiguous().detach()
            c = onetime_infos[1].transpose(0, 1).contiguous().detach()
            self.cells = (h, c)
        advantages, returns = self._gae_and_return(obs, obs_next, rewards, dones)
        advantages, returns = advantages.detach(), returns.detach()
        trace['advantages'] = advantages.numpy().copy()
        trace['returns'] = returns.numpy().copy()
        trace['values_raw'] = self.last_values_raw.numpy().copy()
        if self.if_rnn_policy:
            eff_len = self.n_step - self.horizon + 1
            behave_pol = pds[:, :eff_len, :].contiguous().detach()
            actions_iter = actions[:, :eff_len, :].contiguous().detach()
        else:
            behave_pol = pds[:, 0, :].contiguous().detach()
            actions_iter = actions[:, 0, :].contiguous().detach()
        obs_iter = {}
        for mod in obs.keys():
            obs_iter[mod] = {}
            for k in obs[mod].keys():
                if self.if_rnn_policy:
                    obs_iter[mod][k] = obs[mod][k][:, :self.n_step - self.horizon + 1, :].contiguous().detach()
                else:
                    obs_iter[mod][k] = obs[mod][k][:, 0, :].contiguous().detach()
        ref_pol = self.ref_target_model.forward_actor(obs_iter, self.cells).detach()
        for ep in range(self.epoch_policy):
            stats = self._policy_update(obs_iter, actions_iter, advantages, behave_pol, ref_pol)
            curr_pol = self.model.forward_actor(obs_iter, self.cells).detach()
            kl = self.pd.kl(ref_pol, curr_pol).mean()
            stats['_pol_kl'] = kl.item()
            trace['policy'].append(dict(stats))
            if kl.item() > self.kl_target * 4:
                break
        self.kl_record.append(stats['_pol_kl'])
        for _ in range(self.epoch_baseline):
            baseline_stats = self._value_update(obs_iter, returns)
            trace['value'].append(dict(baseline_stats))
        for k in baseline_stats:
            stats[k] = baseline_stats[k]
        behave_likelihood = self.pd.likelihood(actions_iter, behave_pol)
        curr_likelihood = self.pd.likelihood(actions_iter, curr_pol)
        stats['_avg_return_targ'] = returns.mean().item()
        stats['_avg_log_sig'] = self.model.p['actor.log_var'].mean().item()
        stats['_avg_behave_likelihood'] = behave_likelihood.mean().item()
        stats['_avg_is_weight'] = (curr_likelihood / (behave_likelihood + 1e-4)).mean().item()
        stats['_ref_behave_diff'] = self.pd.kl(ref_pol, behave_pol).mean().item()
        stats['_lr'] = self.actor_optim.param_groups[0]['lr']
        if self.use_z_filter:
            self.model.z_update(obs_iter)
            stats['obs_running_mean'] = float(np.mean(self.model.z_filter.running_mean()))
            stats['obs_running_square'] = float(np.mean(self.model.z_filter.running_square()))
            stats['obs_running_std'] = float(np.mean(self.model.z_filter.running_std()))
        if self.use_r_filter:                                    # ppo.py:583-584
            stats['reward_mean'] = self.reward_filter.reward_mean()
        self.trace = trace
        return stats

    def learn(self, batch):                                      # ppo.py:588-613
        b = self._preprocess_batch_ppo(batch)
        stats = self._optimize(b['obs'], b['actions'], b['rewards'], b['obs_next'],
                               b['persistent_infos'], b.get('onetime_infos'), b['dones'])
        self.exp_counter += self.batch_size
        return stats

    def _post_publish(self):                                     # ppo.py:637-666
        final_kl = np.mean(self.kl_record)
        if self.ppo_mode == 'clip':
            if final_kl > self.kl_target * self.adjust_threshold[1]:
                if self.clip_range[0] < self.clip_epsilon:
                    self.clip_epsilon = self.clip_epsilon / self.clip_scale
            elif final_kl < self.kl_target * self.adjust_threshold[0]:
                if self.clip_range[1] > self.clip_epsilon:
                    self.clip_epsilon = self.clip_epsilon * self.clip_scale
        else:
            if final_kl > self.kl_target * self.adjust_threshold[1]:
                if self.beta_range[1] > self.beta:
                    self.beta = self.beta * self.adapt_scale
            elif final_kl < self.kl_target * self.adjust_threshold[0]:
                if self.beta_range[0] < self.beta:
                    self.beta = self.beta / self.adapt_scale
        self.ref_target_model.load_from(self.model)
        self.kl_record = []
        self.exp_counter = 0
