"""Small host-side helpers the plugin contract relies on."""
import collections
import time


class AutoInitializeMeta(type):
    """Calls ``obj._initialize()`` after ``__init__`` has fully run (including subclasses'),
    the construction contract of the reference's Agent / Learner / Replay plugins
    (surreal/utils/common.py:232-275)."""

    def __call__(cls, *args, **kwargs):
        obj = super().__call__(*args, **kwargs)
        if not hasattr(obj, '_initialize'):
            raise AssertionError('AutoInitializeMeta requires that subclass implements '
                                 '_initialize()')
        obj._initialize()
        return obj


class AttrDict(dict):
    """recursive attribute-access dict; what the reference's learner sees as a ``BeneDict``
    batch (learner/base.py:10, ppo.py:597-605: ``batch.obs``, ``batch.actions`` ...)"""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class TimedTracker(object):
    """track_increment() is True at most once per `interval` seconds
    (what gates parameter publishing, learner/base.py:90-93,133-134)"""

    def __init__(self, interval):
        self.interval = interval
        self._last = time.time()

    def track_increment(self):
        now = time.time()
        if now - self._last >= self.interval:
            self._last = now
            return True
        return False


class MovingAverageRecorder(object):
    """exponentially weighted mean with bias correction (surreal/utils/common.py:466-491):
    cum = decay * cum + value, norm = decay * norm + 1, value = cum / norm (0 before any record)"""

    def __init__(self, decay=0.95):
        self.decay = decay
        self.cum_value = 0
        self.normalization = 0

    def add_value(self, value):
        self.cum_value = self.cum_value * self.decay + value
        self.normalization = self.normalization * self.decay + 1
        return self.cum_value / self.normalization

    def cur_value(self):
        return 0 if self.normalization == 0 else self.cum_value / self.normalization


class PeriodicScalars(object):
    """The throttle in front of a scalar sink (surreal/session/tracker.py:107-166,
    ``PeriodicTensorplex`` with is_average=True, keep_full_history=False): every ``add_scalars`` call
    is remembered per tag (the last ``period`` values), and every ``period``-th call forwards the
    per-tag means with the call count as the global step."""

    def __init__(self, sink, period):
        assert isinstance(period, int) and period > 0
        self.sink, self.period = sink, period
        self.calls = 0
        self._history = {}

    def add_scalars(self, tag_value_dict, global_step=None):
        for tag, value in tag_value_dict.items():
            self._history.setdefault(tag, collections.deque(maxlen=self.period)).append(value)
        self.calls += 1
        if self.calls % self.period:
            return None
        current = {tag: sum(h) / len(h) for tag, h in self._history.items()}
        if self.sink is not None:
            self.sink.add_scalars(current, self.calls if global_step is None else global_step)
        return current
