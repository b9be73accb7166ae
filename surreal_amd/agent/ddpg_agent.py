"""
DDPGAgent (surreal/agent/ddpg_agent.py:26-200): deterministic actor + exploration noise.
``act(obs)`` is the batch-1 reference contract; ``act_batch(obs)`` evaluates all actors of a GPU
in one actor forward (HIP).  Exploration scale follows the reference: sigma_i =
max_sigma * agent_id / num_agents (max_sigma / 3 for a single agent), ddpg_agent.py:78-84.
Parameter-space noise (param_noise.py: 'normal' / 'adaptive_normal') perturbs the parameters an
agent fetches, as ddpg_agent.py:136-153, 174-175.
"""
import collections
import copy
import time

import numpy as np
import torch

from surreal_amd import kernels as KN
from surreal_amd.env import ExpSenderWrapperSSARNStepBootstrap
from surreal_amd.model.ddpg_net import DDPGModel
from surreal_amd.session import ConfigError
from .action_noise import NormalActionNoise, OrnsteinUhlenbeckActionNoise
from .param_noise import NormalParameterNoise, AdaptiveNormalParameterNoise
from .base import Agent


class DDPGAgent(Agent):
    def __init__(self, learner_config, env_config, session_config, agent_id, agent_mode,
                 render=False):
        super().__init__(learner_config=learner_config, env_config=env_config,
                         session_config=session_config, agent_id=agent_id, agent_mode=agent_mode,
                         render=render)
        self.action_dim = self.env_config.action_spec.dim[0]
        self.obs_spec = self.env_config.obs_spec
        self.sleep_time = self.env_config.get('sleep_time', 0.0)
        self.frame_stack_concatenate_on_env = self.env_config.get('frame_stack_concatenate_on_env', True)
        ex = self.learner_config.algo.exploration
        self.param_noise = None                              # ddpg_agent.py:68-72
        self.param_noise_type = ex.param_noise_type
        self.param_noise_sigma = ex.param_noise_sigma
        self.param_noise_alpha = ex.param_noise_alpha
        self.param_noise_target_stddev = ex.param_noise_target_stddev
        self.noise_type = ex.noise_type
        n_agents = self.env_config.get('num_agents', 1)
        if n_agents == 1:
            self.sigma = ex.max_sigma / 3.0
        else:
            self.sigma = ex.max_sigma * (float(agent_id) / n_agents)
        self.K = KN.default_kernels()
        self.device = KN.default_device()
        conv = self.learner_config.model.get('conv_spec', None) or {}
        self.model = DDPGModel(obs_spec=self.obs_spec, action_dim=self.action_dim,
                               use_layernorm=self.learner_config.model.use_layernorm,
                               actor_fc_hidden_sizes=self.learner_config.model.actor_fc_hidden_sizes,
                               critic_fc_hidden_sizes=self.learner_config.model.critic_fc_hidden_sizes,
                               conv_out_channels=conv.get('out_channels'),
                               conv_kernel_sizes=conv.get('kernel_sizes'), conv_strides=conv.get('strides'),
                               conv_hidden_dim=conv.get('hidden_output_dim'),
                               device=self.device, kernels=self.K)
        self.sink = None
        self._init_noise()

    def _init_noise(self):
        self.noise = None
        if self.agent_mode in ['eval_deterministic', 'eval_deterministic_local']:
            return
        if self.noise_type == 'normal':
            self.noise = NormalActionNoise(np.zeros(self.action_dim),
                                           np.ones(self.action_dim) * self.sigma)
        elif self.noise_type == 'ou_noise':
            ex = self.learner_config.algo.exploration
            self.noise = OrnsteinUhlenbeckActionNoise(mu=np.zeros(self.action_dim), sigma=self.sigma,
                                                      theta=ex.theta, dt=ex.dt)
        else:
            raise ConfigError('Noise type {} undefined.'.format(self.noise_type))
        if self.param_noise_type == 'normal':                # ddpg_agent.py:136-147
            self.param_noise = NormalParameterNoise(self.param_noise_sigma)
        elif self.param_noise_type == 'adaptive_normal':
            from surreal_amd.distributed import ModuleDict
            clean = copy.deepcopy(self.model)
            self.param_noise = AdaptiveNormalParameterNoise(
                clean, ModuleDict(self.module_dict(clean)), self.param_noise_target_stddev,
                alpha=self.param_noise_alpha, sigma=self.param_noise_sigma)

    def on_parameter_fetched(self, params, info):         # ddpg_agent.py:149-153
        params = super().on_parameter_fetched(params, info)
        if self.param_noise:
            if any(torch.is_tensor(v) for sd in params.values() for v in sd.values()):
                # the in-process hand-off passes device snapshots: noise acts on their wire form
                # ({module: {name: ndarray}}, module_dict.py:34-45)
                params = {name: collections.OrderedDict(
                    (k, v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                    for k, v in sd.items()) for name, sd in params.items()}
            params = self.param_noise.apply(params)
        return params

    def act(self, obs):                                   # ddpg_agent.py:155-184
        if self.sleep_time > 0.0:
            time.sleep(self.sleep_time)
        obs_t = collections.OrderedDict()                  # ddpg_agent.py:166-172 (frames stay uint8 here)
        for modality in obs:
            obs_t[modality] = collections.OrderedDict()
            for key in obs[modality]:
                v = np.asarray(obs[modality][key])
                if modality == 'pixel' and not self.frame_stack_concatenate_on_env and v.ndim == 4:
                    v = np.concatenate(list(v), axis=0)    # a list of stacked frames (ddpg_agent.py:159-165)
                t = torch.as_tensor(v) if modality == 'pixel' else torch.as_tensor(v, dtype=torch.float32)
                obs_t[modality][key] = t.unsqueeze(0).to(self.device)
        action_t = self.model.forward_actor(self.model.forward_perception(obs_t))
        if self.param_noise and self.param_noise_type == 'adaptive_normal':
            self.param_noise.compute_action_distance(obs_t, action_t)
        action = action_t.cpu().numpy()[0]
        action = action.clip(-1, 1)
        if self.agent_mode not in ['eval_deterministic', 'eval_deterministic_local']:
            action += self.noise()          # in place: the fp64 draw is rounded into the fp32 action (:180)
        return action.clip(-1, 1)

    def act_batch(self, obs, sigmas=None, eps=None, generator=None):
        """obs [n, D] on the device -> actions [n, A]; sigmas [n] per-actor exploration scale
        (default: this agent's sigma for all rows)"""
        a = self.model.forward_actor(obs).clamp_(-1.0, 1.0)
        if self.agent_mode not in ['eval_deterministic', 'eval_deterministic_local']:
            if eps is None:
                eps = torch.randn(a.shape, device=a.device, generator=generator)
            s = self.sigma if sigmas is None else sigmas.view(-1, 1)
            a = a + eps * s
        return a.clamp_(-1.0, 1.0)

    def module_dict(self, model=None):
        return {'ddpg': self.model if model is None else model}

    def pre_episode(self):                                # ddpg_agent.py:205-208
        super().pre_episode()
        if self.agent_mode not in ['eval_deterministic', 'eval_deterministic_local']:
            self.noise.reset()                            # the OU process restarts with every episode

    def set_experience_sink(self, sink):
        self.sink = sink

    def prepare_env_agent(self, env):
        env = super().prepare_env_agent(env)
        return ExpSenderWrapperSSARNStepBootstrap(env, self.learner_config, self.session_config,
                                                  sink=self.sink)
