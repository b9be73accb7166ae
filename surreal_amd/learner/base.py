"""
Learner plugin base class: same constructor, hooks and main loop as the reference's
``surreal.learner.base.Learner`` (surreal/learner/base.py:21-389), with the ZeroMQ /
tensorplex process plumbing replaced by in-process hand-off:

  reference                                   here
  ------------------------------------------  ---------------------------------------------
  LearnerDataPrefetcher (worker processes     ``attach_replay(replay)`` or
  REQ -> replay ZmqServer, base.py:102-110)   ``set_data_source(fn)``; ``fetch_batch()`` pulls
                                              ``batch_size`` experiences, runs
                                              ``_prefetcher_preprocess`` then ``preprocess``
  ParameterPublisher (ZMQ PUB -> parameter    ``add_parameter_listener(fn)``: called with
  servers -> agents, base.py:90-100)          (module_dict(), info) on publish -- agents in the
                                              same process share device parameters
  tensorplex / loggerplex clients             ``self.tensorplex`` = in-memory scalar recorder,
  (base.py:159-175)                           ``self.log`` = std logging

Out of scope: sockets, shards, cluster launchers.
"""
import collections.abc
import logging
import os
import pickle
import time

import numpy as np
import torch

from surreal_amd.session import Config
from surreal_amd.utils import AutoInitializeMeta, AttrDict, TimedTracker


class ScalarRecorder(object):
    """stands where the throttled tensorplex client stood (learner/base.py:176-190):
    keeps the last value and a short history of every scalar group.  A group may be a mapping that
    is still being read back from the device (DeferredStats): it is stored untouched and only
    looked at when somebody asks for ``latest`` / iterates ``history``."""

    def __init__(self, keep=1000):
        self.keep = keep
        self.history = []
        self._latest = {}
        self._merged = 0            # history entries already folded into _latest

    def add_scalars(self, scalars, global_step=None):
        self.history.append((global_step, scalars if isinstance(scalars, DeferredStats) else dict(scalars)))
        drop = len(self.history) - self.keep
        if drop > 0:
            # fold ONLY what is about to be dropped: newer entries may be read-backs still in
            # flight, and looking at them would turn every learn() into a device -> host sync
            for _, old in self.history[self._merged:drop]:
                self._latest.update(old)
            del self.history[:drop]
            self._merged = max(self._merged - drop, 0)

    @property
    def latest(self):
        for _, scalars in self.history[self._merged:]:
            self._latest.update(scalars)
        self._merged = len(self.history)
        return self._latest


class DeferredStats(collections.abc.Mapping):
    """the statistics of a learn() whose device -> host read-back is still in flight: a read-only
    mapping that waits for the copy the first time it is looked at.  The learner resolves it at the
    latest when the next learn() has been enqueued, so the host decodes step k while the GPU runs
    step k + 1 instead of the GPU idling for the read-back and the Python in between."""

    def __init__(self, resolve):
        self._resolve, self._value = resolve, None

    def _get(self):
        if self._value is None:
            self._resolve()
        return self._value

    def __getitem__(self, k):
        return self._get()[k]

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())

    def __repr__(self):
        return 'DeferredStats(%r)' % (self._value if self._value is not None else '<in flight>')

    def __reduce__(self):
        # crossing a process boundary (pickle, multiprocessing queues): the resolved values travel as
        # a plain dict -- never the learner behind the resolver
        return (dict, (dict(self._get()),))

    def copy(self):
        return dict(self._get())


class Learner(metaclass=AutoInitializeMeta):
    def __init__(self, learner_config, env_config, session_config):
        self.learner_config = learner_config
        self.env_config = env_config
        self.session_config = session_config
        self.current_iter = 0
        self._data_source = None
        self._parameter_listeners = []
        self._setup_logging()
        self._setup_checkpoint()

    # ---- abstract (base.py:46-85) ---------------------------------------------------------
    def learn(self, batch_exp):
        raise NotImplementedError

    def module_dict(self):
        raise NotImplementedError

    def checkpoint_attributes(self):
        return []

    # ---- AutoInitializeMeta hook (base.py:117-128) ----------------------------------------
    def _initialize(self):
        if self.session_config.checkpoint.restore:
            self.restore_checkpoint()
        self._setup_publish()

    # ---- parameter publish (base.py:90-144) -----------------------------------------------
    def _setup_publish(self):
        interval = self.learner_config.parameter_publish.min_publish_interval
        self._ps_publish_tracker = TimedTracker(interval)

    def add_parameter_listener(self, fn):
        self._parameter_listeners.append(fn)

    def should_publish_parameter(self):
        return self._ps_publish_tracker.track_increment()

    def attach_parameter_publisher(self, publish_fn):
        """also publish serialised parameter blobs (ModuleDict.dumps + info with the content hash)
        through ``publish_fn((binary, info))`` -- e.g. ``ParameterServer.set_storage`` or a socket"""
        from surreal_amd.distributed import ParameterPublisher
        self._ps_publisher = ParameterPublisher(publish_fn, self.module_dict())

    def _publish(self, iteration, message=''):
        self._publish_seq = getattr(self, '_publish_seq', 0) + 1
        info = {'time': time.time(), 'iteration': iteration, 'message': message,
                'publish_seq': self._publish_seq}
        for fn in self._parameter_listeners:
            fn(self.module_dict(), info)
        if getattr(self, '_ps_publisher', None) is not None:
            self._ps_publisher.publish(iteration, message)

    def publish_parameter(self, iteration, message=''):
        self._publish(iteration, message)

    # ---- data (base.py:102-115,149-154) ---------------------------------------------------
    def set_data_source(self, fn):
        """fn() -> list of `batch_size` experiences (what a replay's sample() returns)"""
        self._data_source = fn

    def attach_replay(self, replay):
        bs = self.learner_config.replay.batch_size

        def pull():
            while not replay.start_sample_condition():
                time.sleep(0.001)
            return replay.sample(bs)
        self._data_source = pull

    def start_prefetching(self, depth=2):
        """Host-fed learners (experiences arriving in host memory): a background thread pulls and aggregates the next
        batch into pinned struct-of-arrays staging and its host-to-device copy runs on a second stream UNDER the
        current learn() (surreal_amd.distributed.LearnerDataPrefetcher + PinnedBatchStager; the reference's
        LearnerDataPrefetcher, learner/base.py:102-110, data_fetcher.py:9-73).  fetch_batch() then hands out
        device-resident batches from `depth` fixed address sets (one captured graph each)."""
        from surreal_amd.distributed.data_fetcher import LearnerDataPrefetcher, PinnedBatchStager
        if self._data_source is None:
            raise RuntimeError('no data source attached: call attach_replay / set_data_source')
        # The staging is sized from the HOST batch and the learner's device-moving preprocess() is NOT run in the worker:
        # the stager's device twins (fp32, uint8 for camera frames) are what preprocess() would make of the batch.
        # (Run there, a preprocess that moves the batch to the device -- DDPG's -- cost H->D->H->D, and its allocator /
        # synchronous copy calls came from the worker thread while the main thread may be capturing a hipGraph.)
        # A subclass whose preprocess() does MORE than that move (scaling, reshapes, key renames -- the reference's
        # documented per-algorithm hook, learner/base.py:102-110 `main_preprocess=self.preprocess`) still gets it:
        # applied on the main thread to the device batch the stager hands out (fetch_batch), so both paths -- with
        # and without prefetching -- deliver the same batches.  The hooks this package ships are marked
        # `device_move_only` and skipped.
        own = type(self).preprocess
        self._prefetch_main_preprocess = None if getattr(own, 'device_move_only', False) else self.preprocess
        first = self._as_attr(self._prefetcher_preprocess(self._data_source()))
        device = getattr(self, 'device', 'cpu')
        stager = PinnedBatchStager(first, depth=depth, device=device)
        self.graph_input_sets = max(getattr(self, 'graph_input_sets', 1), depth)
        src = self._data_source
        self._prefetch_queue = LearnerDataPrefetcher(self.session_config, self.learner_config.replay.batch_size,
                                                     worker_preprocess=self._prefetcher_preprocess,
                                                     main_preprocess=None,
                                                     source=lambda bs: src(), stager=stager)
        # the batch that sized the staging is not lost: it goes through slot 0 first
        stager.stage(first)
        self._prefetch_queue.preprocess_queue.put(('slot', 0))
        self._prefetch_queue.start()
        return self._prefetch_queue

    @staticmethod
    def _as_attr(data):
        return AttrDict(data) if isinstance(data, dict) and not isinstance(data, AttrDict) else data

    def fetch_batch(self):
        if getattr(self, '_prefetch_queue', None) is not None:
            t0 = time.time()
            data = self._prefetch_queue.get()
            if getattr(self, '_prefetch_main_preprocess', None) is not None:
                data = self._prefetch_main_preprocess(data)
            self.fetch_time_s = time.time() - t0
            return data
        if self._data_source is None:
            raise RuntimeError('no data source attached: call attach_replay / set_data_source')
        t0 = time.time()
        data = self._prefetcher_preprocess(self._data_source())
        if isinstance(data, dict) and not isinstance(data, AttrDict):
            data = AttrDict(data)
        data = self.preprocess(data)
        self.fetch_time_s = time.time() - t0
        return data

    def fetch_iterator(self):
        while True:
            yield self.fetch_batch()

    def preprocess(self, batch):
        """the per-algorithm hook on a fetched batch (reference: Learner.preprocess, learner/base.py:149-154).  With
        prefetching on it receives the DEVICE-resident batch of the staging slot (leaves are torch tensors); an
        override whose only job is the host-to-device move sets `preprocess.device_move_only = True` and is skipped
        there (the stager has already done it)."""
        return batch
    preprocess.device_move_only = True

    def _prefetcher_preprocess(self, batch):
        return batch

    # ---- logging / metrics (base.py:159-255) ------------------------------------------------
    def _setup_logging(self):
        self.log = logging.getLogger('surreal_amd.learner')
        self.tensorplex = ScalarRecorder()
        self.init_time = time.time()
        self.learn_time_s = 0.0
        self.fetch_time_s = 0.0
        self.publish_time_s = 0.0
        self.iter_time_s = 0.0
        self.last_time = self.init_time
        self.last_iter = 0

    def generate_tensorplex_report(self):
        """the reference's .core/* and .system/* gauges (base.py:201-255); exp_per_s is its own
        learner-ingest metric (sub-trajectories/s; x n_step = env-steps/s)"""
        now = time.time()
        iters = self.current_iter - self.last_iter
        elapsed = max(now - self.last_time, 1e-9)
        self.last_iter, self.last_time = self.current_iter, now
        iter_time = self.iter_time_s + 1e-6
        iter_per_s = iters / elapsed
        m = {
            '.core/learn_time_s': self.learn_time_s + 1e-6,
            '.core/fetch_time_s': self.fetch_time_s + 1e-6,
            '.core/publish_time_s': self.publish_time_s + 1e-6,
            '.core/iter_time_s': iter_time,
            '.system/iter_per_s': iter_per_s,
            '.system/exp_per_s': iter_per_s * self.learner_config.replay.batch_size,
            '.system/compute_load_percent': min(self.learn_time_s / iter_time * 100, 100),
            '.system/io_fetch_experience_load_percent': min(self.fetch_time_s / iter_time * 100, 100),
            '.system/io_publish_load_percent': min(self.publish_time_s / iter_time * 100, 100),
        }
        self.tensorplex.add_scalars(m)
        return m

    # ---- checkpoint (base.py:260-313) in the reference's on-disk format (utils/checkpoint.py) ----
    def _setup_checkpoint(self):
        from surreal_amd.utils.checkpoint import PeriodicCheckpoint
        tracked_attrs = self.checkpoint_attributes()
        assert isinstance(tracked_attrs, (list, tuple)), \
            'checkpoint_attributes must return a list of string attr names'
        ck = self.session_config.checkpoint.learner
        self._periodic_checkpoint = PeriodicCheckpoint(
            os.path.join(self.session_config.folder, 'checkpoint'), name='learner',
            period=ck.periodic, min_interval=ck.min_interval, tracked_obj=self,
            tracked_attrs=list(tracked_attrs), keep_history=ck.keep_history, keep_best=ck.keep_best)

    def save_checkpoint(self, global_steps, score=None, **info):
        """unconditional save (the reference only exposes the periodic one)"""
        from surreal_amd.utils.checkpoint import Checkpoint
        Checkpoint.save(self._periodic_checkpoint, score=score, global_steps=global_steps, **info)
        return self._periodic_checkpoint.ckpt_path(global_steps)

    def periodic_checkpoint(self, global_steps, score=None, **info):
        return self._periodic_checkpoint.save(score=score, global_steps=global_steps,
                                              reload_metadata=False, **info)

    def restore_checkpoint(self):
        SC = self.session_config
        restore_folder = SC.checkpoint.restore_folder
        if restore_folder and os.path.basename(os.path.normpath(restore_folder)) != 'checkpoint':
            restore_folder = os.path.join(restore_folder, 'checkpoint')   # base.py:301-304
        restored = self._periodic_checkpoint.restore(
            target=SC.checkpoint.learner.restore_target, mode=SC.checkpoint.learner.mode,
            reload_metadata=True, check_ckpt_exists=True, restore_folder=restore_folder)
        return bool(restored)

    # ---- main loop (base.py:348-389) -----------------------------------------------------------
    def main(self):
        self.main_setup()
        while True:
            self.main_loop()

    def main_setup(self):
        self.save_config()
        self._iter_t0 = time.time()
        self.publish_parameter(0, message='batch ' + str(0))

    def main_loop(self):
        data = self.fetch_batch()
        t0 = time.time()
        self.learn(data)
        self.learn_time_s = time.time() - t0
        if self.should_publish_parameter():
            t1 = time.time()
            self.publish_parameter(self.current_iter, message='batch ' + str(self.current_iter))
            self.publish_time_s = time.time() - t1
        now = time.time()
        self.iter_time_s = now - getattr(self, '_iter_t0', now)
        self._iter_t0 = now
        self.current_iter += 1

    def save_config(self):
        folder = self.session_config.folder
        os.makedirs(folder, exist_ok=True)
        Config(learner_config=self.learner_config, env_config=self.env_config,
               session_config=self.session_config).dump_file(os.path.join(folder, 'config.yml'))
