from .base import Agent, AGENT_MODES, PeriodicTracker
from .ppo_agent import PPOAgent
