"""
Experience windowing on the host: turn one agent's step stream into the experiences the learner
consumes.  The emission rules are the reference's (surreal/env/exp_sender_wrapper.py:31-264) --
pinned by tests/golden/hostpath.json, recorded from the reference's own wrappers -- but they are
stated here the way the device tier states them (csrc/smx_replay.hip ``smx_window_emit_f32``:
``dst[a*W + w, j] = src[a, start + w*advance + j]``): as index arithmetic over the steps of an
episode, held in a ring of ``n_step`` slots, instead of a queue that is filled, scanned and popped.

  step t of an episode (t = 0, 1, ...) lives in slot t % n_step
  PPO sub-trajectory w  = steps [w*advance, w*advance + n_step), emitted right after its last step
                          was recorded; advance = min(stride, n_step) (the reference pops `stride`
                          entries off a queue that holds only n_step); an episode of T steps yields
                          ``windows_per_episode(T, n_step, stride)`` of them, a partial tail is dropped
  DDPG n-step transition j = (s_j, a_j, R_j, s_{j+n}, done_{j+n-1}), emitted after step j + n - 1

The ZeroMQ ``ExpSender`` (hash-dedup + pyarrow + PUSH socket, surreal/distributed/exp_sender.py:10-98)
is replaced by any callable sink -- normally ``replay._insert_wrapper`` of a replay in the same
process, or ``surreal_amd.distributed.ExpSender.send_exp`` for the reference's wire format.
"""
import copy

from surreal_amd.session import Config, ConfigError, BASE_SESSION_CONFIG, BASE_LEARNER_CONFIG
from .base import Wrapper


def window_advance(n_step, stride):
    """steps between the starts of consecutive sub-trajectories of one episode"""
    return min(int(stride), int(n_step))


def windows_per_episode(T, n_step, stride):
    """how many sub-trajectories an episode of T steps yields (0 when T < n_step): the W the device
    tier passes to smx_window_emit_f32 for the same rollout"""
    if T < n_step:
        return 0
    return (T - n_step) // window_advance(n_step, stride) + 1


class ExpSenderWrapperBase(Wrapper):
    def __init__(self, env, learner_config, session_config, sink=None):
        super().__init__(env)
        self.session_config = Config(session_config).extend(BASE_SESSION_CONFIG)
        self.learner_config = Config(learner_config).extend(BASE_LEARNER_CONFIG)
        self.sink = sink
        self.sent = 0

    def set_sink(self, sink):
        self.sink = sink

    def _emit(self, exp):
        if self.sink is None:
            raise RuntimeError('no experience sink attached (pass sink=replay._insert_wrapper)')
        self.sent += 1
        self.sink(exp)


class ExpSenderWrapperSSAR(ExpSenderWrapperBase):
    """{'obs': [s, s'], 'action', 'reward', 'done', 'info'} per step (exp_sender_wrapper.py:31-67)"""

    def __init__(self, env, learner_config, session_config, sink=None):
        super().__init__(env, learner_config, session_config, sink)
        self._obs = None

    def _reset(self):
        self._obs, info = self.env.reset()
        return self._obs, info

    def _step(self, action):
        obs_next, reward, done, info = self.env.step(action)
        self.send([[self._obs, obs_next], action, reward, done, info])
        self._obs = obs_next
        return obs_next, reward, done, info

    def send(self, data):
        obs_array, action, reward, done, info = data
        self._emit({'obs': obs_array, 'action': action, 'reward': reward, 'done': done, 'info': info})


class ExpSenderWrapperSSARNStepBootstrap(ExpSenderWrapperSSAR):
    """DDPG's n-step transitions (exp_sender_wrapper.py:72-112).

    Transition j (opened at step j) collects the rewards of steps j .. j + n - 1 and is emitted with
    the observation and the done flag of step j + n - 1.  The reference weights the reward of a later
    step t by gamma ** (n - i - 1), i being the transition's position in its queue WHEN r_t arrives:
    that is gamma ** (t - j) once the queue is full (t >= n - 1), but gamma ** (n - 1 - j) while it
    still fills up at the start of an episode -- the first n - 1 transitions of every episode are
    mis-discounted.  Kept as is, stated as that exponent rule.  The last
    n - 1 transitions of an episode are never emitted (reset starts a new episode's slots)."""

    def __init__(self, env, learner_config, session_config, sink=None):
        super().__init__(env, learner_config, session_config, sink)
        self.n_step = self.learner_config.algo.n_step
        self.gamma = self.learner_config.algo.gamma
        self._slots = [None] * self.n_step          # transition j -> slot j % n_step
        self._t = 0                                 # steps taken in the current episode

    def _reset(self):
        self._obs, info = self.env.reset()
        self._t = 0
        return self._obs, info

    def _discount_exponent(self, j, t):
        return (t - j) if t >= self.n_step - 1 else (self.n_step - 1 - j)

    def _step(self, action):
        n, t = self.n_step, self._t
        obs_next, reward, done, info = self.env.step(action)
        for j in range(max(0, t - n + 1), t):       # the transitions still open, oldest first
            tr = self._slots[j % n]
            tr[0][1] = obs_next
            tr[2] += pow(self.gamma, self._discount_exponent(j, t)) * reward
            tr[3] = done
        self._slots[t % n] = [[self._obs, obs_next], action, reward, done, info]
        if t >= n - 1:                              # transition t - n + 1 has seen its n rewards
            self.send(self._slots[(t - n + 1) % n])
        self._t = t + 1
        self._obs = obs_next
        return obs_next, reward, done, info


class ExpSenderWrapperMultiStepMovingWindowWithInfo(ExpSenderWrapperBase):
    """PPO's sub-trajectories (exp_sender_wrapper.py:153-264).  ``action`` is the training-mode return
    of PPOAgent.act: (action_choice, [onetime_infos, persistent_infos]); a window carries the
    onetime infos (the LSTM state) of its FIRST step and the persistent infos (the policy's
    distribution) of every step, and the observation that follows its last step as ``obs_next``."""

    FIELDS = ('obs', 'actions', 'rewards', 'dones', 'onetime', 'persistent_infos', 'infos')

    def __init__(self, env, learner_config, session_config, sink=None):
        super().__init__(env, learner_config, session_config, sink)
        self.n_step = self.learner_config.algo.n_step
        self.stride = self.learner_config.algo.stride
        if self.stride < 1:
            raise ConfigError('stride {} for experience generation cannot be less than 1'
                              .format(self.learner_config.algo.stride))
        self.advance = window_advance(self.n_step, self.stride)
        self._ring = {k: [None] * self.n_step for k in self.FIELDS}     # step t -> slot t % n_step
        self._t = 0
        self._ob = None

    def _reset(self):
        obs, info = self.env.reset()
        self._ob = copy.deepcopy(obs)         # a snapshot: the simulator may reuse its buffers
        self._t = 0                           # windows never cross episodes
        return self._ob, info

    def _step(self, action):
        action_choice, action_info = action
        obs_next, reward, done, info = self.env.step(action_choice)
        t, n = self._t, self.n_step
        slot = t % n
        for k, v in zip(self.FIELDS, (self._ob, action_choice, reward, done, action_info[0], action_info[1], info)):
            self._ring[k][slot] = v
        start = t + 1 - n                     # the window that ends with this step, if one starts there
        if start >= 0 and start % self.advance == 0:
            self.send(start, obs_next)
        self._t = t + 1
        self._ob = copy.deepcopy(obs_next)
        return obs_next, reward, done, info

    def _column(self, key, start):
        ring, n = self._ring[key], self.n_step
        return [ring[(start + j) % n] for j in range(n)]

    def send(self, start, obs_next):
        self._emit({'obs': self._column('obs', start), 'obs_next': obs_next,
                    'actions': self._column('actions', start),
                    'onetime_infos': self._ring['onetime'][start % self.n_step],
                    'persistent_infos': self._column('persistent_infos', start),
                    'rewards': self._column('rewards', start), 'dones': self._column('dones', start),
                    'infos': self._column('infos', start), 'n_step': self.n_step})
