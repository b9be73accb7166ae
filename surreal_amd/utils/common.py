"""Small host-side helpers the plugin contract relies on."""
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
