"""
Experience windowing wrappers: turn an agent's step stream into the experiences the learner
consumes.  Same emission rules as the reference (surreal/env/exp_sender_wrapper.py:72-264);
the ZeroMQ ``ExpSender`` (hash-dedup + pyarrow + PUSH socket, surreal/distributed/
exp_sender.py:10-98) is replaced by any callable sink -- normally ``replay._insert_wrapper`` of
a replay living in the same process.
"""
import copy
from collections import deque

from surreal_amd.session import Config, ConfigError, BASE_SESSION_CONFIG, BASE_LEARNER_CONFIG
from .base import Wrapper


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
    """{'obs': [s, s'], 'action', 'reward', 'done', 'info'}  (exp_sender_wrapper.py:31-67)"""

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
        self._emit({'obs': obs_array, 'action': action, 'reward': reward, 'done': done,
                    'info': info})


class ExpSenderWrapperSSARNStepBootstrap(ExpSenderWrapperSSAR):
    """DDPG's n-step transition builder (exp_sender_wrapper.py:72-112).  Quirks kept on purpose
    (SURVEY.md Appendix A.5 / E): a queued entry at position i receives the new reward weighted
    gamma^(n_step - i - 1) -- wrong during the ramp-up of an episode -- and the last n_step - 1
    transitions of an episode are never sent (the queue is cleared on reset)."""

    def __init__(self, env, learner_config, session_config, sink=None):
        super().__init__(env, learner_config, session_config, sink)
        self.n_step = self.learner_config.algo.n_step
        self.gamma = self.learner_config.algo.gamma
        self.last_n = deque()

    def _reset(self):
        self._obs, info = self.env.reset()
        self.last_n.clear()
        return self._obs, info

    def _step(self, action):
        obs_next, reward, done, info = self.env.step(action)
        for i, exp_list in enumerate(self.last_n):
            exp_list[0][1] = obs_next
            exp_list[2] += pow(self.gamma, self.n_step - i - 1) * reward
            exp_list[3] = done
        self.last_n.append([[self._obs, obs_next], action, reward, done, info])
        if len(self.last_n) == self.n_step:
            self.send(self.last_n.popleft())
        self._obs = obs_next
        return obs_next, reward, done, info


class ExpSenderWrapperMultiStepMovingWindowWithInfo(ExpSenderWrapperBase):
    """PPO's sub-trajectory builder (exp_sender_wrapper.py:153-264): a window of `n_step`
    transitions is emitted as soon as it is full, then `stride` entries are popped; windows
    never cross episodes (reset clears the queue; a partial tail is dropped).  `action` is the
    training-mode return of PPOAgent.act: (action_choice, [onetime_infos, persistent_infos])."""

    def __init__(self, env, learner_config, session_config, sink=None):
        super().__init__(env, learner_config, session_config, sink)
        self._ob = None
        self.n_step = self.learner_config.algo.n_step
        self.stride = self.learner_config.algo.stride
        if self.stride < 1:
            raise ConfigError('stride {} for experience generation cannot be less than 1'
                              .format(self.learner_config.algo.stride))
        self.last_n = deque()

    def _reset(self):
        obs, info = self.env.reset()
        self._ob = copy.deepcopy(obs)
        self.last_n.clear()
        return self._ob, info

    def _step(self, action):
        action_choice, action_info = action
        obs_next, reward, done, info = self.env.step(action_choice)
        self.last_n.append([self._ob, action_choice, reward, done, action_info[0], action_info[1],
                            info])
        if len(self.last_n) == self.n_step:
            self.send(self.last_n, obs_next)
            for _ in range(self.stride):
                if len(self.last_n) > 0:
                    self.last_n.popleft()
        self._ob = copy.deepcopy(obs_next)
        return obs_next, reward, done, info

    def send(self, data, obs_next):
        obs, actions, rewards, dones, persistent_infos, infos = [], [], [], [], [], []
        onetime_infos = None
        for ob, action, reward, done, onetime_info, persistent_info, info in data:
            obs.append(ob)
            actions.append(action)
            rewards.append(reward)
            dones.append(done)
            infos.append(info)
            persistent_infos.append(persistent_info)
            if onetime_infos is None:
                onetime_infos = onetime_info
        self._emit({'obs': obs, 'obs_next': obs_next, 'actions': actions,
                    'onetime_infos': onetime_infos, 'persistent_infos': persistent_infos,
                    'rewards': rewards, 'dones': dones, 'infos': infos, 'n_step': len(data)})
