"""
Adapters between simulator APIs and Surreal's environment protocol, and the observation
transforms stacked on top of them (surreal/env/wrapper.py:165-513, surreal/env/make_env.py:41-136).

The simulators themselves (Gym / MuJoCo, Robosuite) are CPU programs outside this repository;
the adapters are duck-typed against their documented surface so that they work with the real
packages when those are installed and with scripted stand-ins in the tests:

  Gym        ``reset() -> obs``, ``step(a) -> (obs, r, done, info)``, ``observation_space.shape``,
             ``action_space.shape``, ``render(mode='rgb_array')``
  Robosuite  ``reset() -> dict``, ``step(a) -> (dict, r, done, info)``, ``observation_spec() -> dict``,
             ``dof``, ``sim.render(...)``

Every transform is an ``ObsTransform``: one function applied to the observation of ``reset`` and
``step`` and one applied to ``observation_spec()``.
"""
import collections
import gc

import numpy as np

from .base import Wrapper, FrameStackWrapper

OD = collections.OrderedDict

SPEC_SURREAL_CLASSIC = 'SURREAL_CLASSIC'      # wrapper.py:12-15 (SpecFormat)
SPEC_DM_CONTROL = 'DM_CONTROL'
SPEC_MUJOCOMANIP = 'MUJOCOMANIP'


class ObsTransform(Wrapper):
    """observation -> observation, spec -> spec; everything else passes through"""
    spec_format = SPEC_SURREAL_CLASSIC

    def map_obs(self, obs):
        return obs

    def map_spec(self, spec):
        return spec

    def _reset(self):
        obs, info = self.env.reset()
        return self.map_obs(obs), info

    def _step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self.map_obs(obs), reward, done, info

    def observation_spec(self):
        return self.map_spec(self.env.observation_spec())


def _is_box(space):
    return hasattr(space, 'shape') and space.shape is not None and not hasattr(space, 'n')


class GymAdapter(Wrapper):
    """wrapper.py:165-213: a Box observation becomes {'low_dim': {'flat_inputs': obs}}"""
    spec_format = SPEC_SURREAL_CLASSIC

    def __init__(self, env, env_config):
        super().__init__(env)
        if env_config.pixel_input:
            raise AssertionError('Pixel input training not supported with OpenAI Gym')

    @staticmethod
    def _low_dim(x):
        return OD([('low_dim', {'flat_inputs': x})])

    def _reset(self):
        return self._low_dim(self.env.reset()), {}

    def _step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self._low_dim(obs), reward, done, info

    def observation_spec(self):
        space = self.env.observation_space
        if not _is_box(space):
            raise ValueError('Discrete observation currently not supported')
        return self._low_dim(space.shape)

    def action_spec(self):
        space = self.env.action_space
        if not _is_box(space):
            raise ValueError('Discrete observation currently not supported')
        return {'type': 'continuous', 'dim': space.shape}

    def _render(self, *args, **kwargs):
        return self.env.render(mode='rgb_array')

    @property
    def unwrapped(self):
        return getattr(self.env, 'unwrapped', self.env)


class RobosuiteWrapper(Wrapper):
    """wrapper.py:216-291: sorts the simulator's flat dict into modalities ('image' is
    'camera0'), repeats actions (reward = mean over the repeats), appends depth as a 4th channel"""
    spec_format = SPEC_MUJOCOMANIP

    def __init__(self, env, env_config):
        env.metadata = {}
        super().__init__(env)
        self.use_depth = bool(env_config.use_depth and env_config.pixel_input)
        self._wanted = env_config.observation
        self._action_repeat = env_config.action_repeat or 1

    def _sort(self, flat, verbose=False):
        pixel, low = OD(), OD()
        for key in flat:
            if key == 'image' and 'camera0' in self._wanted['pixel']:
                pixel['camera0'] = flat[key]
            elif key in self._wanted['pixel']:
                pixel[key] = flat[key]
            elif key in self._wanted['low_dim']:
                low[key] = flat[key]
            elif verbose:
                print('Mujoco: skipping observation key:', key)
        out = OD()
        if pixel:
            out['pixel'] = pixel
        if low:
            out['low_dim'] = low
        return out

    def _with_depth(self, d):
        if self.use_depth:
            d['image'] = np.concatenate((d['image'], np.expand_dims(d['depth'], 2)), 2)
        return d

    def _step(self, action):
        rewards = []
        for _ in range(self._action_repeat):
            obs, reward, done, info = self.env.step(action)
            rewards.append(reward)
            if done:
                break
        return self._sort(self._with_depth(obs)), np.mean(rewards), done, info

    def _reset(self):
        return self._sort(self._with_depth(self.env.reset())), {}

    def observation_spec(self):
        spec = self._with_depth(self.env.observation_spec())
        for k in spec:
            spec[k] = tuple(np.array(spec[k]).shape)
        return self._sort(spec, verbose=True)

    def action_spec(self):
        return {'dim': (self.env.dof,), 'type': 'continuous'}

    def _render(self, *args, **kwargs):
        return self.env.sim.render(camera_name='frontview', height=512, width=512, depth=False)

    @property
    def unwrapped(self):
        return self.env


class DMControlAdapter(Wrapper):
    """dm_wrapper.py:35-88: dm_control TimeSteps -> (obs, reward, done, info).  Low-dim tasks put
    the task's observation dict under 'low_dim'; pixel tasks (dm_control's pixels.Wrapper) rename
    'pixels' to pixel/camera0.  A None reward (first step) counts as 0; done = the LAST step."""
    spec_format = SPEC_DM_CONTROL

    def __init__(self, env, is_pixel_input):
        env.metadata = {}
        super().__init__(env)
        self.is_pixel_input = is_pixel_input

    def _modality(self, obs):
        if self.is_pixel_input:
            return OD([('pixel', OD([('camera0', obs['pixels'])]))])
        return OD([('low_dim', obs)])

    @staticmethod
    def _is_last(ts):
        last = getattr(ts, 'last', None)
        return bool(last()) if callable(last) else int(ts.step_type) == 2      # StepType.LAST

    def _step(self, action):
        ts = self.env.step(action)
        reward = ts.reward
        if reward is None:
            print('None reward')
            reward = 0
        return self._modality(ts.observation), reward, self._is_last(ts), {}

    def _reset(self):
        return self._modality(self.env.reset().observation), {}

    def observation_spec(self):
        out = OD()
        for modality, entries in self._modality(self.env.observation_spec()).items():
            out[modality] = OD((key, spec.shape) for key, spec in entries.items())
        return out

    def action_spec(self):
        return {'type': 'continuous', 'dim': self.env.action_spec().shape}

    @property
    def unwrapped(self):
        return self.env


def wrap_dm_control(task_env, env_config):
    """the wrapper stack of make_env.py:125-135 around an already loaded dm_control task"""
    pix = env_config.pixel_input
    env = DMControlAdapter(task_env, pix)
    env = FilterWrapper(env, env_config)
    env = ObservationConcatenationWrapper(env)
    if pix:
        env = TransposeWrapper(env)
        env = GrayscaleWrapper(env)
        if env_config.frame_stacks > 1:
            env = FrameStackWrapper(env, env_config)
    return _finish(env, env_config)


class FilterWrapper(ObsTransform):
    """keeps only the modalities / keys listed in env_config.observation (wrapper.py:474-513)"""

    def __init__(self, env, env_config):
        super().__init__(env)
        self._allowed = env_config.observation

    def _keep(self, tree, verbose=False):
        out = OD()
        for modality in tree:
            if modality not in self._allowed:
                continue
            kept = OD()
            for key in tree[modality]:
                if key in self._allowed[modality]:
                    kept[key] = tree[modality][key]
                elif verbose:
                    print('Skipping observation key:', modality, '/', key)
            out[modality] = kept
        return out

    def map_obs(self, obs):
        return self._keep(obs)

    def map_spec(self, spec):
        return self._keep(spec, verbose=True)


class ObservationConcatenationWrapper(ObsTransform):
    """all 'low_dim' entries concatenated, in order, into one vector (wrapper.py:294-333)"""

    def __init__(self, env, concatenated_obs_name='flat_inputs'):
        super().__init__(env)
        self._name = concatenated_obs_name

    def map_obs(self, obs):
        if 'low_dim' in obs:
            parts = list(obs['low_dim'].values())
            if parts:
                del obs['low_dim']                       # re-inserted last, like the reference
                obs['low_dim'] = OD([(self._name, np.concatenate(parts))])
        return obs

    def map_spec(self, spec):
        if 'low_dim' in spec:
            total = 0
            for shape in spec['low_dim'].values():
                assert len(shape) == 1
                total += shape[0]
            spec['low_dim'] = OD([(self._name, (total,))])
        return spec


class TransposeWrapper(ObsTransform):
    """camera frames (H, W, C) -> (C, H, W) (wrapper.py:336-363)"""

    def map_obs(self, obs):
        for key in obs.get('pixel', ()):
            obs['pixel'][key] = obs['pixel'][key].transpose((2, 0, 1))
        return obs

    def map_spec(self, spec):
        for key in spec.get('pixel', ()):
            H, W, C = spec['pixel'][key]
            spec['pixel'][key] = (C, H, W)
        return spec


class GrayscaleWrapper(ObsTransform):
    """RGB (3, H, W) -> (1, H, W): channel mean accumulated AND stored in uint8
    (``np.mean(x, 0, 'uint8')``, wrapper.py:370-377 -- the sum wraps modulo 256)"""

    def map_obs(self, obs):
        for key in obs['pixel']:
            frame = obs['pixel'][key]
            C, H, W = frame.shape
            assert C == 3
            obs['pixel'][key] = np.mean(frame, 0, 'uint8').reshape(1, H, W)
        return obs

    def map_spec(self, spec):
        for key in spec['pixel']:
            C, H, W = spec['pixel'][key]
            assert C == 3
            spec['pixel'][key] = (1, H, W)
        return spec


# ---------------------------------------------------------------------------------------------
# make_env (make_env.py:41-136): "<category>:<name>"
# ---------------------------------------------------------------------------------------------
def _finish(env, env_config):
    env_config.action_spec = env.action_spec()
    env_config.obs_spec = env.observation_spec()
    return env, env_config


def wrap_gym(gym_env, env_config):
    return _finish(GymAdapter(gym_env, env_config), env_config)


def wrap_robosuite(sim_env, env_config):
    """the wrapper stack of make_env.py:93-104 around an already constructed simulator"""
    env = RobosuiteWrapper(sim_env, env_config)
    env = FilterWrapper(env, env_config)
    env = ObservationConcatenationWrapper(env)
    if env_config.pixel_input:
        env = TransposeWrapper(env)
        if env_config.use_grayscale:
            env = GrayscaleWrapper(env)
        if env_config.frame_stacks:
            env = FrameStackWrapper(env, env_config)
    return _finish(env, env_config)


def make_env(env_config, mode=None):
    """-> (env, env_config with action_spec / obs_spec filled in).  Categories: 'gym' and
    'robosuite' need those packages; 'synthetic:<D>x<A>' is this repository's scripted env."""
    category, name = env_config.env_name.split(':')
    if mode == 'eval' and 'eval_mode' in env_config:
        for k, v in env_config.eval_mode.items():
            env_config[k] = v
    if category == 'gym':
        import gym
        return wrap_gym(gym.make(name), env_config)
    if category == 'robosuite':
        import robosuite
        pix = env_config.pixel_input
        sim = robosuite.make(name, has_renderer=env_config.render, ignore_done=True, use_camera_obs=pix,
                             has_offscreen_renderer=pix, camera_height=84, camera_width=84,
                             render_collision_mesh=False, render_visual_mesh=True, camera_name='agentview',
                             use_object_obs=(not pix), camera_depth=env_config.use_depth, reward_shaping=True)
        return wrap_robosuite(sim, env_config)
    if category == 'dm_control':
        from dm_control import suite
        domain, task = name.split('-')
        task_env = suite.load(domain_name=domain, task_name=task)
        if env_config.pixel_input:
            from dm_control.suite.wrappers import pixels
            task_env = pixels.Wrapper(task_env, render_kwargs={'height': 84, 'width': 84, 'camera_id': 0})
        return wrap_dm_control(task_env, env_config)
    if category == 'synthetic':
        from .synthetic_env import SyntheticEnv
        D, A = (int(v) for v in name.split('x'))
        env = SyntheticEnv(D, A, episode_len=env_config.get('limit_episode_length', 0) or 200)
        return _finish(env, env_config)
    raise ValueError('Unknown environment category: {}'.format(category))


def make_env_config(env_config, mode=None):
    """make_env.py:16-38: builds the env only to read its specs, then drops it before any fork"""
    env, env_config = make_env(env_config, mode)
    del env
    gc.collect()
    return env_config
