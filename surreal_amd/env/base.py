"""
Environment protocol of the reference (surreal/env/base.py:36-140, surreal/env/wrapper.py:18-162):
``reset() -> (obs, info)``, ``step(action) -> (obs, reward, done, info)``, observations are
``OrderedDict[modality][key] -> np.ndarray`` (docs/env.md:48-77), subclasses override the
underscored methods.  The simulators (Gym / Robosuite / dm_control) need MuJoCo and run on CPU; their
adapters and the observation transforms live in adapters.py, the episode monitors in monitor.py.
"""
import collections

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


def stack_sources(step, n_stack, first=0):
    """Which raw frames make up the stacked observation of `step` (FrameStackWrapper, surreal/env/wrapper.py:407-472):
    the last `n_stack` frames, oldest first, where the frames "before" the episode's first one are that first frame
    (a reset fills the history with it).  The single statement of the rule: the host wrapper below and the device
    gather (smx_frame_stack_u8 over a rollout's raw frames) both evaluate it."""
    return [max(step - (n_stack - 1) + i, first) for i in range(n_stack)]


class MaxStepWrapper(Wrapper):
    """an episode ends after at most `max_steps` steps (wrapper.py:142-162): a budget that reset() refills and
    every step() draws on; `done` is forced once it is spent"""

    def __init__(self, env, max_steps):
        super().__init__(env)
        if max_steps <= 0:
            raise ValueError('MaxStepWrapper received max_steps')
        self.max_steps = max_steps
        self._left = max_steps

    @property
    def current_step(self):
        return self.max_steps - self._left

    def _reset(self):
        self._left = self.max_steps
        return self.env.reset()

    def _step(self, action):
        self._left -= 1
        observation, reward, done, info = self.env.step(action)
        return observation, reward, bool(done) or self._left <= 0, info


class FrameStackWrapper(Wrapper):
    """"obs stacking" for pixel observations (wrapper.py:407-472): the observation carries the last
    `frame_stacks` camera frames -- concatenated on the channel axis, or as a list when
    ``frame_stack_concatenate_on_env`` is off.  Stated as index arithmetic: the episode's raw observations
    are kept by step number (only the last `n` are retained) and the stacked observation of step j is
    frames ``stack_sources(j, n)``; the device tier runs the same rule as a gather over the rollout's raw
    frames (SyntheticVecEnv, smx_frame_stack_u8)."""

    def __init__(self, env, env_config):
        super().__init__(env)
        self.n = env_config.frame_stacks
        self.frame_stack_concatenate_on_env = env_config.frame_stack_concatenate_on_env
        self._raw = {}               # step number within the episode -> raw observation (last n kept)
        self._j = 0

    def _observe(self, obs):
        self._raw[self._j] = obs
        self._raw.pop(self._j - self.n, None)
        picked = [self._raw[f] for f in stack_sources(self._j, self.n)]
        pixels = collections.OrderedDict()
        for key in obs['pixel']:
            frames = [h['pixel'][key] for h in picked]
            pixels[key] = np.concatenate(frames, axis=0) if self.frame_stack_concatenate_on_env else frames
        return collections.OrderedDict((k, pixels if k == 'pixel' else v) for k, v in obs.items())

    def _step(self, action):
        obs_next, reward, done, info = self.env.step(action)
        self._j += 1
        return self._observe(obs_next), reward, done, info

    def _reset(self):
        obs, info = self.env.reset()
        self._raw, self._j = {}, 0
        return self._observe(obs), info

    def observation_spec(self):
        spec = self.env.observation_spec()
        for key, (C, H, W) in list(spec.get('pixel', {}).items()):
            spec['pixel'][key] = (C * self.n, H, W)
        return spec
