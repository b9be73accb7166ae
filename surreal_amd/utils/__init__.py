from .common import AutoInitializeMeta, AttrDict, TimedTracker, MovingAverageRecorder, PeriodicScalars
