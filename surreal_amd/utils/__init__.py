from .common import AutoInitializeMeta, AttrDict, TimedTracker
