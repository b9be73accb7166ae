"""
The learner's input side when batches come from HOST memory (experiences sent by CPU agents through the
collector and a host-tier replay): ``LearnerDataPrefetcher`` keeps the reference's interface
(surreal/distributed/data_fetcher.py:9-73: ``session_config, batch_size, worker_preprocess,
main_preprocess``; ``start()``, ``get()``, ``timer``) -- a background thread pulls ``batch_size`` experiences from
the data source, runs the two preprocess hooks (``worker_preprocess`` = the aggregator, in the reference's worker
processes; ``main_preprocess`` = ``Learner.preprocess``) and queues the result.

What is new is where the batch lands.  A 1024 x 128 x 376 batch is 226 MB: over PCIe (~63 GB/s) that is 3.6 ms, three
times the 1.2 ms the learn itself takes -- so the copy of batch k + 1 has to run UNDER learn(k), and it can only do
that from pinned memory on its own stream.  ``PinnedBatchStager`` owns `depth` staging slots, each a set of pinned
host buffers and their device twins in the batch's struct-of-arrays layout; ``stage()`` copies a batch into a slot's
pinned buffers and enqueues the host-to-device copies on the copy stream; ``acquire()`` makes the compute stream wait
for that slot's copy and hands out the DEVICE batch (same nested structure, stable addresses per slot, so the
learner's captured graphs are reused); ``release()`` marks the slot consumed.  Producers that can write in place
(``MultistepAggregatorWithInfo.aggregate(..., out=stager.host_views(slot))``: experience chunks unpacked straight into
the pinned SoA buffers) skip the host-side copy altogether.
"""
import queue
import threading
import time

import numpy as np
import torch


def _leaves(tree, path=()):
    """(path, leaf) pairs of a nested dict / list batch; None leaves are kept (they carry structure)"""
    if isinstance(tree, dict):
        for k, v in tree.items():
            yield from _leaves(v, path + (k,))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            yield from _leaves(v, path + (i,))
    else:
        yield path, tree


def _rebuild(tree, fn, path=()):
    if isinstance(tree, dict):
        return type(tree)((k, _rebuild(v, fn, path + (k,))) for k, v in tree.items())
    if isinstance(tree, (list, tuple)):
        return [_rebuild(v, fn, path + (i,)) for i, v in enumerate(tree)]
    return fn(path, tree)


class PinnedBatchStager(object):
    """`depth` staging slots for batches shaped like `example` (a nested dict of numpy arrays / tensors; uint8 leaves
    stay uint8, everything else travels as float32 -- what the learners' preprocess would make of it)."""

    def __init__(self, example, depth=2, device=None):
        self.device = torch.device(device if device is not None else 'cuda')
        self.on_gpu = self.device.type == 'cuda'
        if self.on_gpu and self.device.index is None:       # (threads set their current device from it)
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.depth = depth
        self.template = example
        self.copy_stream = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.slots = []
        self.bytes_per_batch = 0
        for _ in range(depth):
            host, dev = {}, {}
            for path, leaf in _leaves(example):
                if leaf is None:
                    continue
                a = leaf.detach().cpu().numpy() if torch.is_tensor(leaf) else np.asarray(leaf)
                dt = torch.uint8 if a.dtype == np.uint8 else torch.float32
                host[path] = torch.empty(a.shape, dtype=dt, pin_memory=self.on_gpu)
                dev[path] = torch.empty(a.shape, dtype=dt, device=self.device) if self.on_gpu else host[path]
            self.slots.append({'host': host, 'dev': dev, 'copied': None, 'consumed': None, 'state': 'free'})
        self.bytes_per_batch = sum(t.numel() * t.element_size() for t in self.slots[0]['host'].values())
        self._next = 0
        self._cv = threading.Condition()          # slot states: free -> staged -> held -> free

    def host_views(self, slot):
        """the slot's pinned buffers as numpy arrays in the batch's structure: an in-place producer's `out`"""
        return _rebuild(self.template, lambda p, leaf: None if leaf is None else self.slots[slot]['host'][p].numpy())

    def _wait_free(self, slot, timeout=None):
        """block (the producer thread) until the consumer has released the slot; then its pinned buffers may be
        rewritten as soon as their last host-to-device copy is done"""
        s = self.slots[slot]
        with self._cv:
            if not self._cv.wait_for(lambda: s['state'] == 'free', timeout):
                raise TimeoutError('staging slot %d was not released' % slot)
            s['state'] = 'filling'
        if s['copied'] is not None:
            s['copied'].synchronize()
        return s

    def submit(self, slot):
        """the slot's pinned buffers hold a new batch: enqueue the host-to-device copies"""
        s = self.slots[slot]
        with self._cv:
            s['state'] = 'staged'
        if not self.on_gpu:
            return slot
        if s['consumed'] is not None:
            self.copy_stream.wait_event(s['consumed'])      # the learn that read the device twins has finished
        with torch.cuda.stream(self.copy_stream):
            for p, h in s['host'].items():
                s['dev'][p].copy_(h, non_blocking=True)
            s['copied'] = torch.cuda.Event()
            s['copied'].record(self.copy_stream)
        return slot

    def stage(self, batch, slot=None):
        """copy `batch` (host arrays) into the next slot's pinned buffers and submit it; returns the slot index"""
        if slot is None:
            slot, self._next = self._next, (self._next + 1) % self.depth
        s = self._wait_free(slot)
        for path, leaf in _leaves(batch):
            if leaf is None:
                continue
            a = leaf.detach().cpu().numpy() if torch.is_tensor(leaf) else np.asarray(leaf)
            np.copyto(s['host'][path].numpy(), a, casting='unsafe')
        return self.submit(slot)

    def begin_fill(self, slot=None):
        """for in-place producers: -> (slot, host_views(slot)) once the slot's pinned buffers are free"""
        if slot is None:
            slot, self._next = self._next, (self._next + 1) % self.depth
        self._wait_free(slot)
        return slot, self.host_views(slot)

    def acquire(self, slot):
        """the DEVICE batch of `slot`; the current stream waits for its copy"""
        s = self.slots[slot]
        with self._cv:
            assert s['state'] == 'staged', 'slot %d holds no staged batch' % slot
            s['state'] = 'held'
        if self.on_gpu and s['copied'] is not None:
            torch.cuda.current_stream().wait_event(s['copied'])
        return _rebuild(self.template, lambda p, leaf: None if leaf is None else s['dev'][p])

    def release(self, slot):
        """call after the consumer (learn) has been enqueued on the current stream: the slot's next copy waits on
        the device for that work, the producer thread may refill the pinned buffers"""
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record()
            self.slots[slot]['consumed'] = ev
        with self._cv:
            self.slots[slot]['state'] = 'free'
            self._cv.notify_all()


class _Timer(object):
    def __init__(self):
        self.total, self.n = 0.0, 0

    @property
    def avg(self):
        return self.total / max(self.n, 1)


class LearnerDataPrefetcher(object):
    def __init__(self, session_config, batch_size, worker_preprocess=None, main_preprocess=None, source=None,
                 stager=None):
        """source(batch_size) -> list of experiences (what a replay's sample() answers); stager: a
        PinnedBatchStager (or None: batches stay on the host and the learner's own preprocess moves them)"""
        lc = session_config.learner
        self.max_fetch_queue = int(lc.get('max_prefetch_queue', 1))
        self.max_preprocess_queue = int(lc.get('max_preprocess_queue', 2))
        self.batch_size = batch_size
        self.worker_preprocess = worker_preprocess
        self.main_preprocess = main_preprocess
        self.source = source
        self.stager = stager
        self.preprocess_queue = queue.Queue(maxsize=self.max_preprocess_queue if stager is None
                                            else max(1, stager.depth - 1))
        self.timer = _Timer()
        self._thread = None
        self._stop = threading.Event()
        self.error = None

    def start(self):
        self._thread = threading.Thread(target=self.run, daemon=True)
        self._thread.start()

    def run(self):
        try:
            if self.stager is not None and self.stager.on_gpu:
                torch.cuda.set_device(self.stager.device)       # (the current device is per thread)
            while not self._stop.is_set():
                data = self.source(self.batch_size)
                slot = None
                if self.stager is not None and getattr(self.worker_preprocess, 'accepts_out', False):
                    # the aggregator writes every field straight into the slot's pinned buffers
                    slot, views = self.stager.begin_fill()
                    filled = data = self.worker_preprocess(data, out=views)
                elif self.worker_preprocess is not None:
                    data = self.worker_preprocess(data)
                if self.main_preprocess is not None:
                    data = self.main_preprocess(data)
                if self.stager is not None:
                    if slot is not None and all(a is b for (_, a), (_, b) in zip(_leaves(data), _leaves(filled))):
                        data = ('slot', self.stager.submit(slot))
                    else:
                        data = ('slot', self.stager.stage(data, slot) if slot is None else self._restage(data, slot))
                while not self._stop.is_set():
                    try:
                        self.preprocess_queue.put(data, timeout=0.05)
                        break
                    except queue.Full:
                        pass
        except Exception as e:          # surfaced by get()
            self.error = e

    def _restage(self, data, slot):
        """main_preprocess replaced the in-place batch: copy its result into the slot that is already ours"""
        st = self.stager
        for path, leaf in _leaves(data):
            if leaf is not None:
                a = leaf.detach().cpu().numpy() if torch.is_tensor(leaf) else np.asarray(leaf)
                np.copyto(st.slots[slot]['host'][path].numpy(), a, casting='unsafe')
        return st.submit(slot)

    def get(self):
        """the next batch: device-resident (its copy already in flight or done) when a stager is attached.  The
        previous batch's slot is released here -- the learner has enqueued its learn() by the time it asks again."""
        t0 = time.time()
        if getattr(self, '_held', None) is not None:
            self.stager.release(self._held)
            self._held = None
        while True:
            if self.error is not None:
                raise self.error
            try:
                data = self.preprocess_queue.get(timeout=0.05)
                break
            except queue.Empty:
                pass
        if isinstance(data, tuple) and len(data) == 2 and data[0] == 'slot':
            self._held = data[1]
            data = self.stager.acquire(data[1])
        self.timer.total += time.time() - t0
        self.timer.n += 1
        return data

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(2.0)
