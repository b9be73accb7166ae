"""
TEST INFRASTRUCTURE ONLY.  CPU restatement (PyTorch fp32) of the reference's DDPG update
(surreal/learner/ddpg.py:244-352, 403-428; surreal/model/ddpg_net.py:13-95;
model_builders/builders.py:35-84; use_layernorm when the parameters carry actor.ln* / critic.ln*: torchx's L.LayerNorm(1)
taken as torch.nn.LayerNorm over the features, as oracle/ref_shims.py does), with every switch of ddpg.py: the TD3 double
critic and action regularisation, camera observations (the perception CNN), LayerNorm -- alone and together.
oracle/gen_golden_ddpg.py pins it bit-for-bit against the reference's own DDPGLearner.

Canonical parameter names: actor.fc{1,2,3}.{W,b} (D->h1->h2->A, tanh);
critic.fc1.{W,b} (D->c1), critic.fc2.{W,b} (c1+A->c2), critic.fc3.{W,b} (c2->1).
"""
import collections

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def make_ddpg_params(D, A, actor_hidden=(300, 200), critic_hidden=(400, 300), seed=3, layernorm=False):
    rs = np.random.RandomState(seed)
    p = collections.OrderedDict()

    def lin(name, out_f, in_f):
        b = 1.0 / np.sqrt(in_f)
        p[name + '.W'] = rs.uniform(-b, b, (out_f, in_f)).astype(np.float32)
        p[name + '.b'] = rs.uniform(-b, b, (out_f,)).astype(np.float32)
    lin('actor.fc1', actor_hidden[0], D)
    lin('actor.fc2', actor_hidden[1], actor_hidden[0])
    lin('actor.fc3', A, actor_hidden[1])
    lin('critic.fc1', critic_hidden[0], D)
    lin('critic.fc2', critic_hidden[1], critic_hidden[0] + A)
    lin('critic.fc3', 1, critic_hidden[1])
    if layernorm:       # affine parameters away from their (1, 0) initial values, so that a test notices them
        for net, hs in (('actor', actor_hidden), ('critic', critic_hidden)):
            for i, h in enumerate(hs):
                p['%s.ln%d.W' % (net, i + 1)] = (1.0 + 0.2 * rs.randn(h)).astype(np.float32)
                p['%s.ln%d.b' % (net, i + 1)] = (0.1 * rs.randn(h)).astype(np.float32)
    return p


def make_ddpg_pixel_params(low_dim, A, pixel, conv_hidden, actor_hidden, critic_hidden, seed=3,
                           channels=(16, 32), kernels=(8, 4), strides=(4, 2), layernorm=False):
    """parameters of the pixel DDPGModel (ddpg_net.py:37-51): CNNStemNetwork perception (builders.py:8-33)
    whose features are concatenated IN FRONT of the low-dim vector (ddpg_net.py:67-78), then the
    MLPs over conv_hidden + low_dim inputs.  cnn.* names as in oracle/ppo_oracle.py."""
    C, H, W = pixel
    p = make_ddpg_params(conv_hidden + low_dim, A, actor_hidden, critic_hidden, seed=seed, layernorm=layernorm)
    rs = np.random.RandomState(seed + 1000)
    h1, w1 = (H - kernels[0]) // strides[0] + 1, (W - kernels[0]) // strides[0] + 1
    h2, w2 = (h1 - kernels[1]) // strides[1] + 1, (w1 - kernels[1]) // strides[1] + 1

    def u(shape, fan):
        b = 1.0 / np.sqrt(fan)
        return rs.uniform(-b, b, shape).astype(np.float32)
    p['cnn.conv1.W'] = u((channels[0], C, kernels[0], kernels[0]), C * kernels[0] ** 2)
    p['cnn.conv1.b'] = u((channels[0],), C * kernels[0] ** 2)
    p['cnn.conv2.W'] = u((channels[1], channels[0], kernels[1], kernels[1]), channels[0] * kernels[1] ** 2)
    p['cnn.conv2.b'] = u((channels[1],), channels[0] * kernels[1] ** 2)
    p['cnn.fc.W'] = u((conv_hidden, channels[1] * h2 * w2), channels[1] * h2 * w2)
    p['cnn.fc.b'] = u((conv_hidden,), channels[1] * h2 * w2)
    return p


class OracleDDPGModel(object):
    def __init__(self, params):
        self.p = collections.OrderedDict(
            (k, torch.tensor(np.asarray(v), dtype=torch.float32).clone().requires_grad_(True))
            for k, v in params.items())

    def actor_params(self):
        return [v for k, v in self.p.items() if k.startswith('actor.')]

    def critic_params(self):
        # the perception CNN trains with the critic (ddpg_net.py:57-61)
        return [v for k, v in self.p.items() if k.startswith('critic.') or k.startswith('cnn.')]

    strides = (4, 2)

    def forward_perception(self, obs):                # ddpg_net.py:67-78, builders.py:8-33
        if not isinstance(obs, dict):
            return obs
        parts = []
        if 'pixel' in obs:
            p = self.p
            x = obs['pixel']['camera0'] / 255.0       # scale_image (ddpg_net.py:90-95)
            x = torch.relu(F.conv2d(x, p['cnn.conv1.W'], p['cnn.conv1.b'], stride=self.strides[0]))
            x = torch.relu(F.conv2d(x, p['cnn.conv2.W'], p['cnn.conv2.b'], stride=self.strides[1]))
            x = torch.relu(F.linear(x.reshape(x.shape[0], -1), p['cnn.fc.W'], p['cnn.fc.b']))
            parts.append(x)
        if 'low_dim' in obs:
            parts.append(obs['low_dim']['flat_inputs'])
        return torch.cat(parts, dim=1)

    def forward_actor(self, x):                       # builders.py:35-56
        p = self.p
        h = self._ln('actor.ln1', torch.relu(F.linear(x, p['actor.fc1.W'], p['actor.fc1.b'])))
        h = self._ln('actor.ln2', torch.relu(F.linear(h, p['actor.fc2.W'], p['actor.fc2.b'])))
        return torch.tanh(F.linear(h, p['actor.fc3.W'], p['actor.fc3.b']))

    def _ln(self, name, h):                           # builders.py:42-48, 65-75 (use_layernorm)
        if name + '.W' not in self.p:
            return h
        return F.layer_norm(h, (h.shape[1],), self.p[name + '.W'], self.p[name + '.b'], 1e-5)

    def forward_critic(self, x, a):                   # builders.py:58-84
        p = self.p
        h = self._ln('critic.ln1', torch.relu(F.linear(x, p['critic.fc1.W'], p['critic.fc1.b'])))
        h = torch.cat((h, a), 1)
        h = self._ln('critic.ln2', torch.relu(F.linear(h, p['critic.fc2.W'], p['critic.fc2.b'])))
        return F.linear(h, p['critic.fc3.W'], p['critic.fc3.b'])

    def load_from(self, other, tau=None):
        with torch.no_grad():
            for k in self.p:
                if tau is None:
                    self.p[k].copy_(other.p[k])                                   # hard_update
                else:
                    self.p[k].copy_(self.p[k] * (1.0 - tau) + other.p[k] * tau)   # soft_update

    def load_critic_from(self, other, tau=None):
        """the second critic's target follows only the critic (ddpg.py:412-415, 422-425)"""
        with torch.no_grad():
            for k in self.p:
                if not (k.startswith('critic.') or k.startswith('cnn.')):     # + perception (ddpg.py:414-415)
                    continue
                if tau is None:
                    self.p[k].copy_(other.p[k])
                else:
                    self.p[k].copy_(self.p[k] * (1.0 - tau) + other.p[k] * tau)

    def numpy_params(self):
        return collections.OrderedDict((k, v.detach().numpy().copy()) for k, v in self.p.items())


class OracleDDPGLearner(object):
    def __init__(self, params, gamma=0.99, n_step=3, lr_actor=1e-4, lr_critic=1e-3,
                 clip_actor_gradient=True, actor_gradient_value_clip=1.0,
                 clip_critic_gradient=False, critic_gradient_value_clip=5.0,
                 actor_regularization=0.0, critic_regularization=0.0,
                 target_update_type='hard', target_update_interval=500, tau=1e-3,
                 use_double_critic=False, use_action_regularization=False, params2=None,
                 batch_size=None):
        self.model = OracleDDPGModel(params)
        self.model_target = OracleDDPGModel(params)
        # TD3 options (ddpg.py:119-147, 162-166): a second critic with its own optimiser and target
        self.use_double_critic = use_double_critic
        self.use_action_regularization = use_action_regularization
        self.batch_size = batch_size
        if use_double_critic:
            self.model2 = OracleDDPGModel(params2)
            self.model_target2 = OracleDDPGModel(params2)
            self.critic_optim2 = torch.optim.Adam(self.model2.critic_params(), lr=lr_critic,
                                                  weight_decay=critic_regularization)
        self.gamma, self.n_step = gamma, n_step
        self.clip_actor_gradient, self.actor_clip = clip_actor_gradient, actor_gradient_value_clip
        self.clip_critic_gradient, self.critic_clip = clip_critic_gradient, critic_gradient_value_clip
        self.critic_optim = torch.optim.Adam(self.model.critic_params(), lr=lr_critic,
                                             weight_decay=critic_regularization)
        self.actor_optim = torch.optim.Adam(self.model.actor_params(), lr=lr_actor,
                                            weight_decay=actor_regularization)
        self.target_update_type = target_update_type
        self.target_update_interval = target_update_interval
        self.tau = tau
        self.target_update_counter = 0
        self.critic_criterion = nn.MSELoss()

    def optimize(self, obs, actions, rewards, obs_next, done):   # ddpg.py:244-352
        m, mt = self.model, self.model_target
        assert actions.max().item() <= 1.0 and actions.min().item() >= -1.0
        obs_next_raw, obs_raw = obs_next, obs
        obs_next = mt.forward_perception(obs_next)               # ddpg_net.py:80-88
        model_policy = mt.forward_actor(obs_next)
        next_Q_target = mt.forward_critic(obs_next, model_policy)
        if self.use_action_regularization:                       # ddpg.py:267-278: AFTER next_Q_target
            noise = np.clip(np.random.normal(0, 0.2, size=(self.batch_size, model_policy.shape[1])), -0.5, 0.5)
            model_policy = model_policy + torch.tensor(noise, dtype=torch.float32)
            model_policy = model_policy.clamp(-1, 1)
        y = rewards + pow(self.gamma, self.n_step) * next_Q_target * (1.0 - done)
        if self.use_double_critic:                               # ddpg.py:280-283
            next_Q_target2 = self.model_target2.forward_critic(
                self.model_target2.forward_perception(obs_next_raw), model_policy)
            y2 = rewards + pow(self.gamma, self.n_step) * next_Q_target2 * (1.0 - done)
            y = torch.min(y, y2)
        y = y.detach()
        obs = m.forward_perception(obs_raw)                      # ddpg.py:287
        y_policy = m.forward_critic(obs, actions.detach())
        y_policy2 = self.model2.forward_critic(self.model2.forward_perception(obs_raw), actions.detach()) \
            if self.use_double_critic else None
        for q in m.critic_params():
            q.grad = None
        critic_loss = self.critic_criterion(y_policy, y)
        critic_loss.backward()
        if self.clip_critic_gradient:
            nn.utils.clip_grad_value_(m.critic_params(), self.critic_clip)
        self.critic_optim.step()
        if self.use_double_critic:                               # ddpg.py:312-319 (critic_loss is overwritten)
            for q in self.model2.critic_params():
                q.grad = None
            critic_loss = self.critic_criterion(y_policy2, y)
            critic_loss.backward()
            if self.clip_critic_gradient:
                nn.utils.clip_grad_value_(self.model2.critic_params(), self.critic_clip)
            self.critic_optim2.step()
        for q in m.actor_params():
            q.grad = None
        actor_loss = -m.forward_critic(obs.detach(), m.forward_actor(obs.detach()))
        actor_loss = actor_loss.mean()
        actor_loss.backward()
        if self.clip_actor_gradient:
            nn.utils.clip_grad_value_(m.actor_params(), self.actor_clip)
        self.actor_optim.step()
        stats = {
            'actor_loss': actor_loss.item(),
            'critic_loss': critic_loss.item(),
            'action_norm': actions.norm(2, 1).mean().item(),
            'rewards': rewards.mean().item(),
            'Q_target': y.mean().item(),
            'Q_policy': y_policy.mean().item(),
        }
        if self.use_double_critic:
            stats['Q_policy2'] = y_policy2.mean().item()
        self._target_update()
        return stats

    def _target_update(self):                                    # ddpg.py:403-428
        if self.target_update_type == 'soft':
            self.model_target.load_from(self.model, self.tau)
            if self.use_double_critic:
                self.model_target2.load_critic_from(self.model2, self.tau)
        else:
            self.target_update_counter += 1
            if self.target_update_counter % self.target_update_interval == 0:
                self.model_target.load_from(self.model)
                if self.use_double_critic:
                    self.model_target2.load_critic_from(self.model2)

    def learn(self, batch):
        t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).clone()  # noqa: E731
        if 'pixel' in batch['obs']:                              # ddpg.py:207-222: uint8 -> float
            conv = lambda o: {'pixel': {'camera0': t(o['pixel']['camera0'])},  # noqa: E731
                              'low_dim': {'flat_inputs': t(o['low_dim']['flat_inputs'])}}
            return self.optimize(conv(batch['obs']), t(batch['actions']), t(batch['rewards']),
                                 conv(batch['obs_next']), t(batch['dones']))
        return self.optimize(t(batch['obs']['low_dim']['flat_inputs']), t(batch['actions']),
                             t(batch['rewards']), t(batch['obs_next']['low_dim']['flat_inputs']),
                             t(batch['dones']))
