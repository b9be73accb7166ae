"""Exploration noise processes for DDPG agents (surreal/agent/action_noise.py:9-39)."""
import numpy as np


class ActionNoise(object):
    def reset(self):
        pass


class NormalActionNoise(ActionNoise):
    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma

    def __call__(self):
        return np.random.normal(self.mu, self.sigma)

    def __repr__(self):
        return 'NormalActionNoise(mu={}, sigma={})'.format(self.mu, self.sigma)


class OrnsteinUhlenbeckActionNoise(ActionNoise):
    def __init__(self, mu, sigma, theta, dt, x0=None):
        self.theta, self.mu, self.sigma, self.dt, self.x0 = theta, mu, sigma, dt, x0
        self.reset()

    def __call__(self):
        x = (self.x_prev + self.theta * (self.mu - self.x_prev) * self.dt
             + self.sigma * np.sqrt(self.dt) * np.random.normal(size=self.mu.shape))
        self.x_prev = x
        return x

    def reset(self):
        self.x_prev = self.x0 if self.x0 is not None else np.zeros_like(self.mu)

    def __repr__(self):
        return 'OrnsteinUhlenbeckActionNoise(mu={}, sigma={})'.format(self.mu, self.sigma)
