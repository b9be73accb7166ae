"""Scripted stand-ins for the simulator APIs the env adapters wrap (Gym, Robosuite), shared by
oracle/gen_golden_envwrap.py (which drives the REFERENCE's wrappers with them) and
tests/test_env_adapters.py (which drives ours): same scripts -> comparable outputs."""
import collections

import numpy as np


class Box(object):
    def __init__(self, shape):
        self.shape = shape


def make_fake_gym(base=object, box=Box):
    class FakeGym(base):
        metadata = {}

        def __init__(self, T=4, D=5, A=2):
            self.T, self.t, self.steps = T, 0, 0
            self.observation_space, self.action_space = box(shape=(D,)), box(shape=(A,))
            self.D = D

        def reset(self):
            self.t = 0
            return np.arange(self.D, dtype=np.float32) * 0.5

        def step(self, action):
            self.t += 1
            self.steps += 1
            obs = np.arange(self.D, dtype=np.float32) + self.t
            return obs, float(np.sum(action)) * self.t, self.t >= self.T, {'t': self.t}

        def render(self, mode='human'):
            return np.full((4, 4, 3), 7 if mode == 'rgb_array' else 0, dtype=np.uint8)

        def close(self):
            self.closed = True
    return FakeGym


class FakeSim(object):
    def render(self, **kw):
        return ('frame', sorted(kw.items()))


class FakeRobosuite(object):
    """reset()/step() return Robosuite's flat dict; 'image' is (H, W, 3) uint8, 'depth' (H, W)"""
    dof = 8

    def __init__(self, T=5, H=6, W=4):
        self.T, self.t, self.steps, self.H, self.W = T, 0, 0, H, W
        self.sim = FakeSim()

    def _obs(self):
        t, H, W = self.t, self.H, self.W
        y, x, c = np.meshgrid(np.arange(H), np.arange(W), np.arange(3), indexing='ij')
        img = ((37 * y + 91 * x + 113 * c + 29 * t) % 256).astype(np.uint8)
        return collections.OrderedDict([
            ('robot-state', np.array([t, 2 * t, -t], dtype=np.float64)),
            ('junk', np.array([9.0, 9.0])),
            ('image', img),
            ('depth', ((y[:, :, 0] + x[:, :, 0] + t) % 7).astype(np.uint8)),
            ('object-state', np.array([0.5, t, t * t, 1.0], dtype=np.float64))])

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, action):
        self.t += 1
        self.steps += 1
        return self._obs(), float(self.t) + float(np.sum(action)), self.t >= self.T, {'t': self.t}

    def observation_spec(self):
        return self._obs()

    def close(self):
        self.closed = True


def to_plain(x):
    """nested observation / spec -> JSON-able structure that keeps key order and dtypes"""
    if isinstance(x, dict):
        return [[k, to_plain(v)] for k, v in x.items()]
    if isinstance(x, np.ndarray):
        return {'dtype': str(x.dtype), 'shape': list(x.shape), 'data': x.reshape(-1).tolist()}
    if isinstance(x, (list, tuple)):
        return [to_plain(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def run_script(env, n_steps, action):
    """reset, then n_steps steps (resetting after a done); returns the JSON-able trace"""
    obs, info = env.reset()
    trace = [{'reset': to_plain(obs), 'info': to_plain(info)}]
    for _ in range(n_steps):
        obs, r, done, info = env.step(action)
        info = {k: v for k, v in info.items() if k != 'episode'}
        trace.append({'obs': to_plain(obs), 'reward': float(r), 'done': bool(done), 'info': to_plain(info)})
        if done:
            obs, info = env.reset()
            trace.append({'reset': to_plain(obs), 'info': to_plain(info)})
    return trace


# ---- dm_control ---------------------------------------------------------------------------------
import enum  # noqa: E402


class StepType(enum.IntEnum):
    FIRST, MID, LAST = 0, 1, 2


TimeStep = collections.namedtuple('TimeStep', 'step_type reward discount observation')


class ArraySpec(object):
    def __init__(self, shape, dtype=np.float64):
        self.shape, self.dtype = shape, np.dtype(dtype)


def make_fake_dm(base=object):
    class FakeDMControl(base):
        """a dm_control task: TimeSteps whose observation is a dict of named vectors (or
        {'pixels': HxWx3} when `pixels` is set, like dm_control.suite.wrappers.pixels.Wrapper)"""

        def __init__(self, T=4, pixels=False):
            self.T, self.t, self.steps, self.pixels = T, 0, 0, pixels

        def _obs(self):
            t = self.t
            if self.pixels:
                y, x, c = np.meshgrid(np.arange(5), np.arange(3), np.arange(3), indexing='ij')
                return collections.OrderedDict(pixels=((11 * y + 50 * x + 90 * c + 17 * t) % 256).astype(np.uint8))
            return collections.OrderedDict([('position', np.array([t, -t, 0.5])), ('junk', np.array([3.0])),
                                            ('velocity', np.array([0.25 * t, 1.0]))])

        def reset(self):
            self.t = 0
            return TimeStep(StepType.FIRST, None, None, self._obs())

        def step(self, action):
            self.t += 1
            self.steps += 1
            last = self.t >= self.T
            reward = None if self.t == 1 else float(np.sum(action)) + self.t
            return TimeStep(StepType.LAST if last else StepType.MID, reward, 1.0, self._obs())

        def observation_spec(self):
            return collections.OrderedDict((k, ArraySpec(v.shape, v.dtype)) for k, v in self._obs().items())

        def action_spec(self):
            return ArraySpec((2,))

        def close(self):
            self.closed = True
    return FakeDMControl


# ---- scripted env for the agent main loop (oracle/gen_golden_agentloop.py, tests/agent_loop_cases.py) ----
def make_scripted_loop_env(base=object):
    """An environment in the reference's protocol (``reset() -> (obs, info)``, ``step(a) -> (obs, r,
    done, info)``) whose observations come from a seeded stream that does NOT depend on the actions
    (so that a 1e-5 difference in an action cannot fork the script), with scripted episode lengths.
    `base` is the Env class of the side under test (the reference's or ours)."""
    class ScriptedLoopEnv(base):
        metadata = {}

        def __init__(self, D, A, episode_lens, seed=0):
            self.D, self.A = D, A
            self.episode_lens = list(episode_lens)
            self.rs = np.random.RandomState(seed)
            self.episode = -1
            self.t = 0
            self.total_steps = 0
            self.actions = []

        def _obs(self):
            return collections.OrderedDict(low_dim=collections.OrderedDict(
                flat_inputs=self.rs.randn(self.D).astype(np.float32)))

        def _reset(self):
            self.episode += 1
            self.t = 0
            return self._obs(), {}

        def _step(self, action):
            self.t += 1
            self.total_steps += 1
            self.actions.append(np.array(action, dtype=np.float64))
            done = self.t >= self.episode_lens[self.episode % len(self.episode_lens)]
            return self._obs(), 0.5 * self.t - 0.125 * self.episode, done, {'t': self.t}

        # (a base without the reset/step -> _reset/_step indirection)
        def reset(self):
            return self._reset()

        def step(self, action):
            return self._step(action)

        def action_spec(self):
            return {'dim': [self.A], 'type': 'continuous'}

        def observation_spec(self):
            return collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=[self.D]))
    return ScriptedLoopEnv
