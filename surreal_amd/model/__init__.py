from .ppo_net import DiagGauss, ZFilter, PPOModel
from .ddpg_net import DDPGModel
