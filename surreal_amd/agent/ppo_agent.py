"""
PPOAgent -- rollout worker with the reference's interface (surreal/agent/ppo_agent.py:16-190).

``act(obs)`` keeps the batch-1 contract (training mode returns ``(action, [onetime_infos,
persistent_infos])`` for the windowing wrapper); ``act_batch(obs)`` is the same computation
for ALL actors of a GPU in one actor forward on the device: mean/std from the policy MLP
(HIP, FP32 MFMA), per-actor exploration noise ``std *= exp(noise_i)`` with
``noise_i ~ U(-log_sig_range, +log_sig_range)`` drawn once per actor (ppo_agent.py:57-61,139),
sample ``a = mean + std * eps``, clip to [-1, 1].
"""
import time
import types

import numpy as np
import torch

from surreal_amd import _lib as L
from surreal_amd import kernels as KN
from surreal_amd.env import ExpSenderWrapperMultiStepMovingWindowWithInfo
from surreal_amd.model.ppo_net import DiagGauss, PPOModel
from .base import Agent


class PPOAgent(Agent):
    def __init__(self, learner_config, env_config, session_config, agent_id, agent_mode,
                 render=False):
        super().__init__(learner_config=learner_config, env_config=env_config,
                         session_config=session_config, agent_id=agent_id, agent_mode=agent_mode,
                         render=render)
        self.action_dim = self.env_config.action_spec.dim[0]
        self.obs_spec = self.env_config.obs_spec
        self.use_z_filter = self.learner_config.algo.use_z_filter
        self.init_log_sig = self.learner_config.algo.consts.init_log_sig
        self.log_sig_range = self.learner_config.algo.consts.log_sig_range
        if self.agent_mode != 'training':                        # ppo_agent.py:49-55
            if self.agent_mode not in ['eval_deterministic_local', 'eval_stochastic_local']:
                self.agent_mode = 'eval_stochastic' if self.env_config.stochastic_eval \
                    else 'eval_deterministic'
        if self.agent_mode != 'training':
            self.noise = 0
        else:
            self.noise = np.random.uniform(low=-self.log_sig_range, high=self.log_sig_range)
        self.rnn_config = self.learner_config.algo.rnn
        self.pd = DiagGauss(self.action_dim)
        self.cells = None
        self.K = KN.default_kernels()
        self.device = KN.default_device()
        self.model = PPOModel(obs_spec=self.obs_spec, action_dim=self.action_dim,
                              model_config=self.learner_config.model, use_cuda=True,
                              init_log_sig=self.init_log_sig, use_z_filter=self.use_z_filter,
                              if_pixel_input=self.env_config.get('pixel_input', False),
                              rnn_config=self.rnn_config, device=self.device, kernels=self.K)
        self.sink = None
        self._batch_noise = None
        self._batch_cells = None
        self._act_ws = None
        self.reset()

    def _zero_cells(self, n):
        """(h, c), each (rnn_layer, n, rnn_hidden) zeros (ppo_agent.py:83-95)"""
        shape = (self.rnn_config.rnn_layer, n, self.rnn_config.rnn_hidden)
        return (torch.zeros(*shape, device=self.device), torch.zeros(*shape, device=self.device))

    # ---- batch-1 reference contract (ppo_agent.py:106-154) ----------------------------------
    def act(self, obs):
        action_info = [[], []]
        obs_tensor = {}
        for mod in obs.keys():
            obs_tensor[mod] = {}
            for k in obs[mod].keys():
                obs_tensor[mod][k] = torch.as_tensor(np.asarray(obs[mod][k]), dtype=torch.float32) \
                    .unsqueeze(0).to(self.device)
        if self.rnn_config.if_rnn_policy:                # ppo_agent.py:133-135
            action_info[0].append(self.cells[0].squeeze(1).cpu().numpy())
            action_info[0].append(self.cells[1].squeeze(1).cpu().numpy())
        action_pd, self.cells = self.model.forward_actor_expose_cells(obs_tensor, self.cells)
        action_pd = action_pd.detach().cpu().numpy()
        action_pd[:, self.action_dim:] *= np.exp(self.noise)
        if self.agent_mode not in ['eval_deterministic', 'eval_deterministic_local']:
            action_choice = self.pd.sample(action_pd)
        else:
            action_choice = self.pd.maxprob(action_pd).copy()
        np.clip(action_choice, -1, 1, out=action_choice)
        action_choice = action_choice.reshape((-1,))
        action_pd = action_pd.reshape((-1,))
        action_info[1].append(action_pd)
        if self.agent_mode != 'training':
            return action_choice
        sleep = self.env_config.get('sleep_time', 0)
        if sleep:
            time.sleep(sleep)
        return action_choice, action_info

    # ---- all actors of a GPU in one forward ---------------------------------------------------
    def act_batch(self, obs, generator=None, eps=None, out_actions=None, out_pd=None):
        """obs [n, D] on the device -> (actions [n, A], pds [n, 2A]) on the device.
        `eps` ([n, A] standard normal) may be injected: exact-parity tests do, and a rollout driver
        draws the whole rollout's noise in one launch.  `out_actions` / `out_pd` (row-strided views
        allowed, e.g. a rollout buffer's slot) receive the results in place.
        Plain-MLP policies take 5 launches per step: z-filter from the running sums, three layers,
        the sampling head (the reference's ~15 ATen ops + a host round trip, ppo_agent.py:106-154)."""
        if isinstance(obs, dict):                    # {'low_dim': {...}, 'pixel': {...}} of [n, ...] tensors
            n = next(iter(next(iter(obs.values())).values())).shape[0]
        else:
            n = obs.shape[0]
        A = self.action_dim
        K = self.K
        if self._batch_noise is None or self._batch_noise.shape[0] != n:
            if self.agent_mode == 'training':
                g = torch.Generator().manual_seed(1234 + int(self.agent_id))
                u = (torch.rand(n, 1, generator=g) * 2 - 1) * self.log_sig_range
            else:
                u = torch.zeros(n, 1)
            self._batch_noise = torch.exp(u).to(self.device)
        if self.rnn_config.if_rnn_policy or self.model.if_pixel:
            # one LSTM step for all n actors; `batch_cells_before` is what every actor's
            # onetime_infos would hold for this step (its state BEFORE acting, :133-135)
            if self.rnn_config.if_rnn_policy:
                if self._batch_cells is None or self._batch_cells[0].shape[1] != n:
                    self._batch_cells = self._zero_cells(n)
                self.batch_cells_before = self._batch_cells
            pd_model, self._batch_cells = self.model.forward_actor_expose_cells(
                obs if isinstance(obs, dict) else {'low_dim': {'flat_inputs': obs}}, self._batch_cells)
            mean = pd_model[:, :A]
        else:
            ws = self._act_ws
            if ws is None or ws.n != n:
                f = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)  # noqa: E731
                net = self.model.actor
                ws = self._act_ws = types.SimpleNamespace(n=n, xn=f(n, net.D), h1=f(n, net.H1),
                                                          h2=f(n, net.H2), mean=f(n, A))
            if self.use_z_filter:
                zf = self.model.z_filter
                K.zfilter_forward_sums(obs, zf.running_sum, zf.running_sumsq, zf.count, zf.eps, ws.xn)
            else:
                ws.xn.copy_(obs)
            K.mlp3_forward(self.model.actor, ws.xn, ws.h1, ws.h2, ws.mean, L.SMX_ACT_TANH)
            mean = ws.mean
        deterministic = self.agent_mode in ['eval_deterministic', 'eval_deterministic_local']
        if not deterministic and eps is None:
            eps = torch.randn(n, A, device=self.device, generator=generator)
        actions = out_actions if out_actions is not None else torch.empty(n, A, device=self.device)
        pd = out_pd if out_pd is not None else torch.empty(n, 2 * A, device=self.device)
        K.diaggauss_sample(mean, self.model.log_var.view(-1), self._batch_noise.view(-1),
                           None if deterministic else eps, actions, pd)
        return actions, pd

    def policy_mean(self, xn, out=None):
        """the policy's mean for observations that are ALREADY z-filtered (plain-MLP policies): the
        three layer launches alone.  A device-resident rollout filters the next observation and
        samples in the environment-step launch (SyntheticVecEnv.rollout)."""
        if self.rnn_config.if_rnn_policy or self.model.if_pixel:
            raise NotImplementedError('policy_mean: plain MLP policies only; use act_batch')
        n = xn.shape[0]
        ws = self._act_ws
        if ws is None or ws.n != n:
            self.act_batch(xn, eps=torch.zeros(n, self.action_dim, device=self.device))   # builds the workspace
            ws = self._act_ws
        out = ws.mean if out is None else out
        self.K.mlp3_forward(self.model.actor, xn, ws.h1, ws.h2, out, L.SMX_ACT_TANH)
        return out

    def batch_noise(self, n):
        """per-actor exploration scale exp(noise_i) [n, 1] (ppo_agent.py:57-61, 139)"""
        if self._batch_noise is None or self._batch_noise.shape[0] != n:
            self.act_batch(torch.zeros(n, self.model.actor.D, device=self.device),
                           eps=torch.zeros(n, self.action_dim, device=self.device))
        return self._batch_noise

    def module_dict(self):
        return {'ppo': self.model}

    def reset(self):                                   # ppo_agent.py:167-183
        """zero LSTM hidden and cell state (start of an episode)"""
        self.cells = self._zero_cells(1) if self.rnn_config.if_rnn_policy else None

    def reset_batch(self, mask=None):
        """act_batch counterpart of reset(): zero the state of the actors in `mask` ([n] bool on
        the device; None = all) whose episodes just ended"""
        if self._batch_cells is None:
            return
        if mask is None:
            self._batch_cells = None
            return
        keep = (~mask).to(torch.float32).view(1, -1, 1)
        self._batch_cells = (self._batch_cells[0] * keep, self._batch_cells[1] * keep)

    def set_experience_sink(self, sink):
        """where windowed experiences go: normally ``replay._insert_wrapper``"""
        self.sink = sink

    def prepare_env_agent(self, env):                  # ppo_agent.py:185-190
        env = super().prepare_env_agent(env)
        return ExpSenderWrapperMultiStepMovingWindowWithInfo(env, self.learner_config,
                                                             self.session_config, sink=self.sink)
