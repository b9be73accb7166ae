"""
Environment protocol of the reference (surreal/env/base.py:36-140, surreal/env/wrapper.py:18-162):
``reset() -> (obs, info)``, ``step(action) -> (obs, reward, done, info)``, observations are
``OrderedDict[modality][key] -> np.ndarray`` (docs/env.md:48-77), subclasses override the
underscored methods.  The simulators (Gym / Robosuite / dm_control) need MuJoCo and run on CPU; their
adapters and the observation transforms live in adapters.py, the episode monitors in monitor.py.
"""
import collections
from collections import deque

import numpy as np


class Env(object):
    metadata = {}

    def _step(self, action):
        raise NotImplementedError

    def _reset(self):
        raise NotImplementedError

    def _render(self, *args, **kwargs):
        pass

    def _close(self):
        pass

    def step(self, action):
        return self._step(action)

    def reset(self):
        return self._reset()

    def render(self, *args, **kwargs):
        return self._render(*args, **kwargs)

    def close(self):
        return self._close()

    def action_spec(self):
        raise NotImplementedError

    def observation_spec(self):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    env = None

    def __init__(self, env):
        self.env = env
        probe = env
        while isinstance(probe, Wrapper):            # wrapper.py:39-50: no double wrapping
            if type(probe).__name__ == type(self).__name__:
                raise RuntimeError('Attempted to double wrap with Wrapper: %s' % type(self).__name__)
            probe = probe.env

    def _step(self, action):
        return self.env.step(action)

    def _reset(self):
        return self.env.reset()

    def _render(self, *args, **kwargs):
        return self.env.render(*args, **kwargs)

    def _close(self):
        return self.env.close()

    def action_spec(self):
        return self.env.action_spec()

    def observation_spec(self):
        return self.env.observation_spec()

    @property
    def unwrapped(self):
        return self.env.unwrapped


class MaxStepWrapper(Wrapper):
    """forces done after `max_steps` steps (wrapper.py:142-162)"""

    def __init__(self, env, max_steps):
        super().__init__(env)
        if max_steps <= 0:
            raise ValueError('MaxStepWrapper received max_steps')
        self.max_steps = max_steps
        self.current_step = 0

    def _reset(self):
        self.current_step = 0
        return self.env.reset()

    def _step(self, action):
        self.current_step += 1
        observation, reward, done, info = self.env.step(action)
        if self.current_step >= self.max_steps:
            done = True
        return observation, reward, done, info


class FrameStackWrapper(Wrapper):
    """"obs stacking" for pixel observations (wrapper.py:407-472): the last `frame_stacks`
    frames concatenated on the channel axis; reset fills the history with the first frame."""

    def __init__(self, env, env_config):
        super().__init__(env)
        self.n = env_config.frame_stacks
        self.frame_stack_concatenate_on_env = env_config.frame_stack_concatenate_on_env
        self._history = deque(maxlen=self.n)

    def _stacked_observation(self, obs):
        pixels = collections.OrderedDict()
        for key in obs['pixel']:
            frames = [h['pixel'][key] for h in self._history]
            pixels[key] = np.concatenate(frames, axis=0) if self.frame_stack_concatenate_on_env \
                else frames
        out = collections.OrderedDict()
        for key in obs:
            out[key] = pixels if key == 'pixel' else obs[key]
        return out

    def _step(self, action):
        obs_next, reward, done, info = self.env.step(action)
        self._history.append(obs_next)
        return self._stacked_observation(obs_next), reward, done, info

    def _reset(self):
        obs, info = self.env.reset()
        for _ in range(self.n):
            self._history.append(obs)
        return self._stacked_observation(obs), info

    def observation_spec(self):
        spec = self.env.observation_spec()
        if 'pixel' in spec:
            for key in spec['pixel']:
                C, H, W = spec['pixel'][key]
                spec['pixel'][key] = (C * self.n, H, W)
        return spec
