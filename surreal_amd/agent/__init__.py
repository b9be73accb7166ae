from .base import Agent, AGENT_MODES, PeriodicTracker
from .ppo_agent import PPOAgent
from .ddpg_agent import DDPGAgent
from .action_noise import NormalActionNoise, OrnsteinUhlenbeckActionNoise
