"""
Exploration noise for the DDPG agents (behaviour of surreal/agent/action_noise.py:9-39).

Both processes are written for a whole rank of actors at once: a draw has the shape of the
location parameter, so ``[A]`` drives one actor (the reference's use) and ``[n, A]`` drives the n
actors of a GPU with one call.  Randomness comes from numpy's global stream, as in the reference.

  gaussian:  x_t = loc + scale * eps_t
  OU:        x_t = x_{t-1} + rate * (loc - x_{t-1}) * dt + scale * sqrt(dt) * eps_t ,  x_{-1} = start or 0
"""
import numpy as np


class _Process(object):
    """stateless by default; ``reset()`` is called at every episode start"""

    kind = 'noise'

    def __init__(self, loc, scale):
        self.mu = np.asarray(loc, dtype=np.float64) if not np.isscalar(loc) else loc
        self.sigma = scale

    def reset(self):
        return None

    def _eps(self):
        return np.random.normal(size=np.shape(self.mu))

    def __repr__(self):
        return '{}(mu={}, sigma={})'.format(type(self).__name__, self.mu, self.sigma)


class NormalActionNoise(_Process):
    kind = 'gaussian'

    def __call__(self):
        return self.mu + self.sigma * self._eps()


class OrnsteinUhlenbeckActionNoise(_Process):
    kind = 'ou'

    def __init__(self, mu, sigma, theta, dt, x0=None):
        super().__init__(mu, sigma)
        self.theta, self.dt, self.x0 = theta, dt, x0
        self._root_dt = float(np.sqrt(dt))
        self.reset()

    def reset(self):
        self.x_prev = np.zeros(np.shape(self.mu)) if self.x0 is None else np.array(self.x0, dtype=np.float64)

    def __call__(self):
        drift = self.theta * (self.mu - self.x_prev) * self.dt
        self.x_prev = self.x_prev + drift + self.sigma * self._root_dt * self._eps()
        return self.x_prev


ActionNoise = _Process
