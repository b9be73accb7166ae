"""
Agent plugin base class: constructor signature, modes, hooks and the episode loop of the
reference's ``surreal.agent.base.Agent`` (surreal/agent/base.py:24-377).  The ZeroMQ parameter
client (REQ 'parameter:<hash>' to a parameter server every ``fetch_parameter_interval`` steps)
is replaced by an in-process hand-off: ``attach_learner(learner)`` subscribes the agent to the
learner's publishes, ``fetch_parameter()`` copies the latest published module state, device to
device -- on one GPU the "parameter server" is a tensor copy.
"""
import logging
import time

from surreal_amd.env import MaxStepWrapper
from surreal_amd.utils import AutoInitializeMeta, MovingAverageRecorder, PeriodicScalars

AGENT_MODES = ['training', 'eval_deterministic', 'eval_stochastic',
               'eval_deterministic_local', 'eval_stochastic_local']


class PeriodicTracker(object):
    """True once every `period` increments (surreal/session/tracker.py:10-79)"""

    def __init__(self, period, init_value=0, init_endpoint=0):
        assert isinstance(period, int) and period > 0
        self.period = period
        self.value = init_value
        self._endpoint = init_endpoint

    def track_increment(self, incr=1):
        self.value += incr
        if self.value - self._endpoint >= self.period:
            self._endpoint += (self.value - self._endpoint) // self.period * self.period
            return True
        return False


class Agent(object, metaclass=AutoInitializeMeta):
    def __init__(self, learner_config, env_config, session_config, agent_id, agent_mode,
                 render=False):
        self.learner_config = learner_config
        self.env_config = env_config
        self.session_config = session_config
        assert agent_mode in AGENT_MODES
        self.agent_mode = agent_mode
        self.agent_id = agent_id
        self.log = logging.getLogger('surreal_amd.agent.%s' % agent_id)
        if self.agent_mode not in ['eval_deterministic_local', 'eval_stochastic_local']:
            self._setup_parameter_pull()
            self._setup_logging()
        self.current_episode = 0
        self.cumulative_steps = 0
        self.current_step = 0
        self.actions_since_param_update = 0
        self.episodes_since_param_update = 0
        self.render = render
        self._published = None       # (module_dict, info) last published by the learner
        self._fetched_iteration = None
        self._env_factory = None

    def _initialize(self):
        self._module_dict = self.module_dict()

    def _setup_parameter_pull(self):
        self._fetch_parameter_mode = self.session_config.agent.fetch_parameter_mode
        self._fetch_parameter_interval = self.session_config.agent.fetch_parameter_interval
        self._fetch_parameter_tracker = PeriodicTracker(self._fetch_parameter_interval)

    def _setup_logging(self):
        """agent/base.py:87-108: the throttled scalar sink ('agent/<id>' or 'eval/<id>') and the
        how-long-was-a-parameter-set-used bookkeeping (weighted over ~100 parameter updates)"""
        from surreal_amd.learner.base import ScalarRecorder
        self.tensorplex_name = '{}/{}'.format('agent' if self.agent_mode == 'training' else 'eval', self.agent_id)
        self.tensorplex = PeriodicScalars(ScalarRecorder(),
                                          self.session_config.tensorplex.update_schedule.agent)
        self.actions_since_param_update = 0
        self.episodes_since_param_update = 0
        self.actions_per_param_update = MovingAverageRecorder(decay=0.99)
        self.episodes_per_param_update = MovingAverageRecorder(decay=0.99)

    # ---- abstract ---------------------------------------------------------------------------
    def act(self, obs):
        raise NotImplementedError

    def module_dict(self):
        raise NotImplementedError

    def reset(self):
        pass

    # ---- parameter hand-off -------------------------------------------------------------------
    def attach_learner(self, learner):
        learner.add_parameter_listener(self._on_publish)
        self._on_publish(learner.module_dict(), {'time': time.time(), 'iteration': 0, 'message': 'initial',
                                                 'publish_seq': 0})

    def _on_publish(self, module_dict, info):
        """keeps a SNAPSHOT of what was published (device copies of the state dicts): an agent that
        fetches later gets the parameters of the publish point -- where PPO synchronised its
        reference policy and beta -- not whatever the learner holds by then.  The snapshot is taken
        once per publish and shared by every agent attached to the learner (Learner._publish hands
        all listeners the same `info`; the copy rides in it)."""
        snap = info.get('_snapshot')
        if snap is None:
            snap = {name: {k: (v.clone() if hasattr(v, 'clone') else v) for k, v in m.state_dict().items()}
                    for name, m in module_dict.items()}
            info['_snapshot'] = snap
        self._published = (snap, info)

    def on_parameter_fetched(self, params, info):      # agent/base.py:160-180
        if self.agent_mode == 'training':
            delay = time.time() - info['time']         # learner -> agent
            self.actions_per_param_update.add_value(self.actions_since_param_update)
            self.episodes_per_param_update.add_value(self.episodes_since_param_update)
            self.tensorplex.add_scalars({
                '.core/parameter_publish_delay_s': delay,
                '.core/actions_per_param_update': self.actions_per_param_update.cur_value(),
                '.core/episodes_per_param_update': self.episodes_per_param_update.cur_value()})
            self.actions_since_param_update = 0
            self.episodes_since_param_update = 0
        return params

    def attach_parameter_client(self, client):
        """fetch through the reference's parameter protocol (surreal_amd.distributed.ParameterClient:
        serialised blobs + content-hash "unchanged" replies) instead of the in-process hand-off"""
        self._param_client = client

    def fetch_parameter(self):
        """agent/base.py:355-363; returns True when new parameters were loaded"""
        client = getattr(self, '_param_client', None)
        if client is not None:
            from surreal_amd.distributed import ModuleDict
            binary, info = client.fetch_parameter_with_info()
            if binary is None:
                return False
            from surreal_amd.utils import serializer
            params = self.on_parameter_fetched(serializer.deserialize(binary), info)
            ModuleDict(self._module_dict).load(params)
            self._fetched_iteration = info.get('iteration')
            return True
        if self._published is None:
            return False
        params, info = self._published
        if info.get('publish_seq', info.get('iteration')) == self._fetched_iteration:
            return False                      # the reference's "hash unchanged" reply
        params = self.on_parameter_fetched(params, info)
        for name, module in self._module_dict.items():
            got = params[name]               # a live module, or its wire form after on_parameter_fetched
            module.load_state_dict(got.state_dict() if hasattr(got, 'state_dict') else got)
        self._fetched_iteration = info.get('publish_seq', info.get('iteration'))
        return True

    def fetch_parameter_info(self):
        return None if self._published is None else self._published[1]

    # ---- hooks (agent/base.py:182-222) -----------------------------------------------------------
    def pre_action(self, obs):
        if self.agent_mode == 'training':
            if self._fetch_parameter_mode == 'step' and self._fetch_parameter_tracker.track_increment():
                self.fetch_parameter()

    def post_action(self, obs, action, obs_next, reward, done, info):
        self.current_step += 1
        self.cumulative_steps += 1
        if self.agent_mode == 'training':
            self.actions_since_param_update += 1
            if done:
                self.episodes_since_param_update += 1

    def pre_episode(self):
        if self.agent_mode == 'training':
            if self._fetch_parameter_mode == 'episode' and self._fetch_parameter_tracker.track_increment():
                self.fetch_parameter()

    def post_episode(self):
        self.current_episode += 1

    # ---- main loops (agent/base.py:224-271) -----------------------------------------------------
    def main(self):
        self.main_setup()
        while True:
            self.main_loop()

    def main_setup(self):
        env = self.get_env()
        self.env = self.prepare_env(env)
        if self.agent_mode == 'training':
            self.fetch_parameter()

    def main_loop(self):
        """one episode"""
        env = self.env
        self.pre_episode()
        obs, info = env.reset()
        total_reward = 0.0
        while True:
            self.pre_action(obs)
            action = self.act(obs)
            obs_next, reward, done, info = env.step(action)
            total_reward += reward
            self.post_action(obs, action, obs_next, reward, done, info)
            obs = obs_next
            if done:
                break
        self.post_episode()
        return total_reward

    def set_env_factory(self, fn):
        """fn() -> Env: overrides make_env (scripted / in-process environments)"""
        self._env_factory = fn

    def get_env(self):
        """agent/base.py:272-281: the env named by env_config.env_name, unless a factory is set"""
        if self._env_factory is not None:
            return self._env_factory()
        from surreal_amd.env import make_env
        mode = 'eval' if self.agent_mode in ('eval_deterministic', 'eval_stochastic') else None
        return make_env(self.env_config, mode=mode)[0]

    def prepare_env(self, env):
        if self.agent_mode == 'training':
            return self.prepare_env_agent(env)
        return self.prepare_env_eval(env)

    def _limit_steps(self, env):
        limit = self.env_config.limit_episode_length      # first: it alters step()'s `done`
        return MaxStepWrapper(env, limit) if limit > 0 else env

    def prepare_env_agent(self, env):                      # agent/base.py:296-312
        from surreal_amd.env import TrainingTensorplexMonitor
        env = TrainingTensorplexMonitor(self._limit_steps(env), agent_id=self.agent_id,
                                        session_config=self.session_config, separate_plots=True)
        self.env_tensorplex = env.tensorplex
        return env

    def prepare_env_eval(self, env):                       # agent/base.py:314-336 (no video recorder)
        env = self._limit_steps(env)
        if self.agent_mode not in ('eval_deterministic_local', 'eval_stochastic_local'):
            from surreal_amd.env import EvalTensorplexMonitor
            env = EvalTensorplexMonitor(env, eval_id=self.agent_id, fetch_parameter=self.fetch_parameter,
                                        session_config=self.session_config)
            self.env_tensorplex = env.tensorplex
        return env

    def main_agent(self):
        self.main()

    def main_eval(self):
        self.main()

    def set_agent_mode(self, agent_mode):
        assert agent_mode in AGENT_MODES
        self.agent_mode = agent_mode
