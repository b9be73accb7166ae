from .ppo_net import DiagGauss, ZFilter, PPOModel
