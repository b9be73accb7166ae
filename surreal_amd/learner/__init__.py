from .base import Learner
from .aggregator import MultistepAggregatorWithInfo, SSARAggregator
from .ppo import PPOLearner
from .ddpg import DDPGLearner
