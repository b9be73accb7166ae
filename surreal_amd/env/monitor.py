"""
Episode bookkeeping around an environment and the periodic reports built on it
(surreal/env/monitor.py:11-218).

``EpisodeMonitor`` records per-episode reward / length / wall time and attaches
``info['episode']`` to the step that ends an episode.  The three reporting monitors differ only in
where a report goes and what happens after it, so they share ``_PeriodicReport``:

  ConsoleMonitor              a small table on stdout every `update_interval` episodes
  TrainingTensorplexMonitor   scalars ':reward' / 'step_per_s' under 'agent/<id>' every
                              tensorplex.update_schedule.training_env episodes
  EvalTensorplexMonitor       the same under 'eval/<id>' (update_schedule.eval_env), then sleeps
                              eval_env_sleep seconds and fetches fresh parameters

The tensorplex process of the reference is replaced by any object with
``add_scalars(dict, global_step=...)`` (default: the in-memory ScalarRecorder the learners use).
"""
import collections
import time

from .base import Wrapper


def _mean(xs):
    return float(sum(xs)) / max(len(xs), 1)


def _fmt(x, precision):
    return ('%.*f' % (precision, x)).rstrip('0').rstrip('.')


class _Every(object):
    """True once per `period` calls (session/tracker.py PeriodicTracker.track_increment)"""

    def __init__(self, period):
        assert isinstance(period, int) and period > 0
        self.period, self.count = period, 0

    def __call__(self):
        self.count += 1
        if self.count >= self.period:
            self.count -= self.period
            return True
        return False


class EpisodeMonitor(Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self._t_first = time.time()
        self._t_episode = None
        self._rewards = None
        self.episode_rewards, self.episode_steps, self.episode_durations = [], [], []
        self.total_steps = 0

    def _reset(self, **kwargs):
        self._rewards = []
        self._t_episode = time.time()
        return self.env.reset(**kwargs)

    def _step(self, action):
        ob, rew, done, info = self.env.step(action)
        self._rewards.append(rew)
        if done:
            now = time.time()
            ep = {'reward': round(sum(self._rewards), 6), 'steps': len(self._rewards),
                  'duration': round(now - self._t_episode, 6),
                  'total_elapsed': round(now - self._t_first, 6)}
            self.episode_rewards.append(ep['reward'])
            self.episode_steps.append(ep['steps'])
            self.episode_durations.append(ep['duration'])
            info['episode'] = ep
        self.total_steps += 1
        if done:
            self._on_episode_end()                   # reports see the finished step counted
        return ob, rew, done, info

    def _on_episode_end(self):
        pass

    @property
    def num_episodes(self):
        return len(self.episode_rewards)

    def step_per_sec(self, average_episodes):
        assert average_episodes > 0
        n = average_episodes
        return sum(self.episode_steps[-n:]) / (sum(self.episode_durations[-n:]) + 1e-7)


class _PeriodicReport(EpisodeMonitor):
    """calls report() at the end of every `period`-th episode"""

    def __init__(self, env, period, window):
        super().__init__(env)
        self._due = _Every(period)
        self._avg = window

    def _on_episode_end(self):
        if self._due():
            self.report(_mean(self.episode_rewards[-self._avg:]), self.step_per_sec(self._avg))

    def report(self, avg_reward, avg_speed):
        raise NotImplementedError


class ConsoleMonitor(_PeriodicReport):
    def __init__(self, env, update_interval=10, average_over=10, extra_rows=None, out=print):
        super().__init__(env, update_interval, average_over)
        if extra_rows is not None and not isinstance(extra_rows, collections.OrderedDict):
            raise AssertionError('extra_rows spec {"row caption": function(total_steps, num_episodes)} '
                                 'must be an OrderedDict')
        self._extra_rows = extra_rows or collections.OrderedDict()
        self._out = out

    def rows(self, avg_reward, avg_speed):
        rows = [['Last {} rewards'.format(self._avg), _fmt(avg_reward, 3)],
                ['Speed iter/s', _fmt(avg_speed, 1)],
                ['Total steps', self.total_steps],
                ['Episodes', self.num_episodes]]
        for caption, fn in self._extra_rows.items():
            rows.append([caption, str(fn(self.total_steps, self.num_episodes))])
        return rows

    def report(self, avg_reward, avg_speed):
        from tabulate import tabulate
        self._out(tabulate(self.rows(avg_reward, avg_speed), tablefmt='simple', numalign='left'))


class _TensorplexReport(_PeriodicReport):
    group = None
    schedule_key = None

    def __init__(self, env, ident, session_config, separate_plots, tensorplex=None):
        period = session_config['tensorplex']['update_schedule'][self.schedule_key]
        super().__init__(env, period, period)
        if tensorplex is None:
            from surreal_amd.learner.base import ScalarRecorder
            tensorplex = ScalarRecorder()
        self.tensorplex = tensorplex
        self.tensorplex_name = '{}/{}'.format(self.group, ident)
        self._separate_plots = separate_plots

    def _tag(self, tag):
        return ':' + tag if self._separate_plots else tag     # tensorplex: ':' = own section

    def report(self, avg_reward, avg_speed):
        self.tensorplex.add_scalars({self._tag('reward'): avg_reward, 'step_per_s': avg_speed},
                                    global_step=self.num_episodes)


class TrainingTensorplexMonitor(_TensorplexReport):
    group, schedule_key = 'agent', 'training_env'

    def __init__(self, env, agent_id, session_config, separate_plots=True, tensorplex=None):
        if not isinstance(agent_id, int):
            raise AssertionError('agent_id must be an int')
        super().__init__(env, agent_id, session_config, separate_plots, tensorplex)


class EvalTensorplexMonitor(_TensorplexReport):
    group, schedule_key = 'eval', 'eval_env'

    def __init__(self, env, eval_id, fetch_parameter, session_config, separate_plots=False,
                 tensorplex=None, sleep=time.sleep):
        super().__init__(env, eval_id, session_config, separate_plots, tensorplex)
        self._throttle_sleep = session_config['tensorplex']['update_schedule']['eval_env_sleep']
        self._sleep = sleep
        self._fetch_parameter = fetch_parameter
        self._fetch_parameter()                      # an evaluator that starts late catches up

    def report(self, avg_reward, avg_speed):
        super().report(avg_reward, avg_speed)
        self._sleep(self._throttle_sleep)
        self._fetch_parameter()
