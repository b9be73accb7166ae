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


# ======================================================================================================================
# Several aggregating PROCESSES behind one staging slot (the reference's `prefetch_processes`,
# surreal/distributed/data_fetcher.py:36-45, learner/base.py:102-110: N worker processes each pull a batch from the
# replay and run `_prefetcher_preprocess` = the aggregator).
#
# What bounds a learner fed by remote CPU agents is not the GPU and not PCIe but the walk over the 131 072 per-step
# Python objects of a 1024 x 128 batch: 27 ms per batch in ONE process (csrc/host/smx_host.c, GIL held for the walk) =
# 4.7e6 env-steps/s, against 3.2e7 the PCIe link carries and 1.2e8 the learner ingests.  The walk cannot be threaded
# (it IS the GIL), so it is spread over processes -- and instead of every worker building a whole batch and shipping
# it to the learner (the reference: 226 MB per batch through a pipe), every worker fills ITS ROWS of the one staging
# slot in place: the slot's host buffers live in POSIX shared memory, registered with the HIP runtime in the learner
# process (hipHostRegister: DMA reads them like hipHostMalloc'ed memory), and the workers receive their experiences from
# their own data source (a connection to the replay / collector of their own, as in the reference -- here a callable
# built inside the worker by `source_factory`).  Per batch the learner process sends W small messages and receives W.
# ======================================================================================================================
import multiprocessing as _mp
from multiprocessing import shared_memory as _shm


def gpu_local_cpus(device):
    """the CPUs of the NUMA node the GPU hangs off (sysfs), or None when that cannot be told: staging memory first
    touched from there is what the GPU's DMA engines read without crossing the socket interconnect"""
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        import os
        cpus &= os.sched_getaffinity(0)
        return cpus or None
    except Exception:
        return None


def _tree_spec(example):
    """[(path, shape, dtype str)] of a batch example: what a worker needs to rebuild numpy views over the shared blocks"""
    spec = []
    for path, leaf in _leaves(example):
        if leaf is None:
            continue
        a = leaf.detach().cpu().numpy() if torch.is_tensor(leaf) else np.asarray(leaf)
        spec.append((path, tuple(a.shape), 'uint8' if a.dtype == np.uint8 else 'float32'))
    return spec


def _views_over(buf, spec):
    """{path: ndarray} over one shared block laid out leaf after leaf (64-byte aligned)"""
    out, off = {}, 0
    for path, shape, dt in spec:
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[path] = np.ndarray(shape, dtype=dt, buffer=buf, offset=off)
        off += (n + 63) & ~63
    return out


def _block_bytes(spec):
    return sum(((int(np.prod(s)) * np.dtype(d).itemsize) + 63) & ~63 for _, s, d in spec)


class SharedBatchStager(PinnedBatchStager):
    """PinnedBatchStager whose host buffers are POSIX shared-memory blocks (one per slot), host-registered for DMA in
    this process and attachable by worker processes (`names`, `spec`)."""

    def __init__(self, example, depth=2, device=None):
        self.device = torch.device(device if device is not None else 'cuda')
        self.on_gpu = self.device.type == 'cuda'
        if self.on_gpu and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.depth = depth
        self.template = example
        self.copy_stream = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.spec = _tree_spec(example)
        nbytes = _block_bytes(self.spec)
        self.blocks, self.names, self.slots, self.registered = [], [], [], []
        # first touch decides which NUMA node a page lives on: the blocks are touched HERE, from the GPU's node
        import os
        self.local_cpus = gpu_local_cpus(self.device) if self.on_gpu else None
        prev_aff = os.sched_getaffinity(0) if self.local_cpus else None
        if self.local_cpus:
            os.sched_setaffinity(0, self.local_cpus)
        for _ in range(depth):
            blk = _shm.SharedMemory(create=True, size=max(nbytes, 64))
            np.ndarray((blk.size,), dtype=np.uint8, buffer=blk.buf)[:] = 0
            self.blocks.append(blk)
            self.names.append(blk.name)
            views = _views_over(blk.buf, self.spec)
            host = {p: torch.from_numpy(v) for p, v in views.items()}
            if self.on_gpu:
                # DMA-able like hipHostMalloc'ed memory: the H2D copies below run asynchronously at the pinned rate
                base = np.ndarray((nbytes,), dtype=np.uint8, buffer=blk.buf).ctypes.data
                rc = torch.cuda.cudart().cudaHostRegister(base, nbytes, 0)
                if int(rc) != 0:
                    raise RuntimeError('hipHostRegister of a %d-byte shared block failed (%r)' % (nbytes, rc))
                self.registered.append(base)
            dev = {p: (torch.empty(h.shape, dtype=h.dtype, device=self.device) if self.on_gpu else h) for p, h in host.items()}
            self.slots.append({'host': host, 'dev': dev, 'copied': None, 'consumed': None, 'state': 'free'})
        if prev_aff:
            os.sched_setaffinity(0, prev_aff)
        self.bytes_per_batch = sum(t.numel() * t.element_size() for t in self.slots[0]['host'].values())
        self._next = 0
        self._cv = threading.Condition()

    def close(self):
        if self.on_gpu:
            torch.cuda.synchronize()
            for base in self.registered:
                torch.cuda.cudart().cudaHostUnregister(base)
        self.registered = []
        for s in self.slots:
            s['host'] = {}
        self.slots = []
        for blk in self.blocks:
            try:
                blk.close()
                blk.unlink()
            except Exception:
                pass
        self.blocks = []


def _plain_structure(tree):
    if isinstance(tree, dict):
        return {k: _plain_structure(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return [_plain_structure(v) for v in tree]
    return None if tree is None else 0


def ppo_aggregate_factory(obs_spec, action_spec):
    """picklable aggregate_factory for AggregationPool: functools.partial(ppo_aggregate_factory, obs_spec, action_spec)"""
    from surreal_amd.learner.aggregator import MultistepAggregatorWithInfo
    return MultistepAggregatorWithInfo(obs_spec, action_spec).aggregate


def _rows(tree, lo, hi):
    """the [lo, hi) rows of every array of a batch-shaped tree of numpy views"""
    if isinstance(tree, dict):
        return type(tree)((k, _rows(v, lo, hi)) for k, v in tree.items())
    if isinstance(tree, (list, tuple)):
        return [_rows(v, lo, hi) for v in tree]
    return None if tree is None else tree[lo:hi]


_CTL_WORDS = 8          # control block (int64): [0] task sequence number, [1] slot, [2] batch size, [3] stop; then per worker:
#                         done sequence number | error flag | seconds in source() (float64 bits) | seconds in aggregate()


def _aggregation_worker(index, world, conn, names, ctl_name, spec, template_spec, source_factory, aggregate_factory,
                        cpus=None):
    """worker process: attach to the slots' shared blocks, then serve tasks posted in the shared control block -- pull
    this worker's share of the batch from its own source and aggregate it straight into its rows of the slot.  Tasks
    and completions travel as single stores to shared memory (a pipe message per worker and batch each way costs the
    dispatching thread ~3 ms per batch at 32 workers -- as much as the aggregation itself); the pipe carries only
    'ready' and tracebacks."""
    blocks, ctl_blk = [], None
    try:
        import signal
        signal.signal(signal.SIGINT, signal.SIG_IGN)
        if cpus:                             # next to the staging memory (and to the GPU): see SharedBatchStager
            import os
            try:
                os.sched_setaffinity(0, cpus)
            except OSError:
                pass
        blocks = [_shm.SharedMemory(name=n) for n in names]
        ctl_blk = _shm.SharedMemory(name=ctl_name)
        ctl = np.ndarray((_CTL_WORDS + 4 * world,), dtype=np.int64, buffer=ctl_blk.buf)
        secs = np.ndarray((_CTL_WORDS + 4 * world,), dtype=np.float64, buffer=ctl_blk.buf)
        views = []
        for blk in blocks:
            flat = _views_over(blk.buf, spec)
            views.append(_rebuild(template_spec, lambda p, leaf: None if leaf is None else flat[p]))
        source = source_factory(index)
        aggregate = aggregate_factory()
        conn.send(('ready', index))
        last, idle = 0, 0
        while True:
            seq = int(ctl[0])
            if ctl[3]:
                break
            if seq == last:
                idle += 1
                time.sleep(0.00005 if idle < 2000 else 0.001)      # (an idle pool backs off to a millisecond)
                continue
            idle = 0
            slot, bs = int(ctl[1]), int(ctl[2])
            lo, hi = bs * index // world, bs * (index + 1) // world
            t0 = time.time()
            exps = source(hi - lo, seq - 1, lo)
            t1 = time.time()
            if hi > lo:
                aggregate(exps, out=_rows(views[slot], lo, hi))
            secs[_CTL_WORDS + 2 * world + index] = t1 - t0
            secs[_CTL_WORDS + 3 * world + index] = time.time() - t1
            last = seq
            ctl[_CTL_WORDS + index] = seq                          # (the completion: one store, after the rows are written)
    except Exception:                       # surfaced by the pool's fill()
        import traceback
        try:
            if ctl_blk is not None:
                np.ndarray((_CTL_WORDS + 4 * world,), dtype=np.int64, buffer=ctl_blk.buf)[_CTL_WORDS + world + index] = 1
            conn.send(('error', index, traceback.format_exc()))
        except Exception:
            pass
    finally:
        views = ctl = secs = None
        for blk in blocks + ([ctl_blk] if ctl_blk is not None else []):
            try:
                blk.close()
            except Exception:
                pass


class AggregationPool(object):
    """`workers` aggregating processes behind a SharedBatchStager.

    source_factory(worker_index) -> source(n, seq, row_lo) -> list of n experiences (runs INSIDE the worker: its own
    connection to the replay / collector; must be picklable, i.e. a module-level callable or functools.partial of one)
    aggregate_factory() -> aggregate(exp_list, out=views) (e.g. functools.partial(ppo_aggregate_factory, obs_spec, action_spec))
    """

    def __init__(self, stager, workers, source_factory, aggregate_factory, start_method='spawn'):
        self.stager = stager
        self.workers = W = int(workers)
        ctx = _mp.get_context(start_method)
        # the template's STRUCTURE only (plain dicts / lists, None leaves kept): workers rebuild it over their views
        tmpl = _plain_structure(stager.template)
        self.ctl_blk = _shm.SharedMemory(create=True, size=8 * (_CTL_WORDS + 4 * W))
        self.ctl = np.ndarray((_CTL_WORDS + 4 * W,), dtype=np.int64, buffer=self.ctl_blk.buf)
        self.secs = np.ndarray((_CTL_WORDS + 4 * W,), dtype=np.float64, buffer=self.ctl_blk.buf)
        self.ctl[:] = 0
        self.conns, self.procs = [], []
        for w in range(W):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_aggregation_worker, args=(w, W, b, stager.names, self.ctl_blk.name, stager.spec, tmpl,
                                                              source_factory, aggregate_factory,
                                                              getattr(stager, 'local_cpus', None)), daemon=True)
            p.start()
            b.close()
            self.conns.append(a)
            self.procs.append(p)
        for c in self.conns:
            if not c.poll(180.0):
                raise TimeoutError('an aggregation worker did not come up within 180 s')
            msg = c.recv()
            if msg[0] != 'ready':
                raise RuntimeError('aggregation worker failed to start: %s' % (msg[-1],))
        self.seq = 0
        self.source_s = self.aggregate_s = 0.0

    def fill(self, slot, batch_size, timeout=120.0):
        """rows [0, batch_size) of `slot`, split evenly over the workers; returns when every worker is done"""
        W = self.workers
        self.seq += 1
        self.ctl[1], self.ctl[2] = slot, batch_size
        self.ctl[0] = self.seq                                 # (posted last: the workers act on the sequence number)
        done, err = self.ctl[_CTL_WORDS:_CTL_WORDS + W], self.ctl[_CTL_WORDS + W:_CTL_WORDS + 2 * W]
        t0 = time.time()
        while not (done == self.seq).all():
            if err.any():
                for c in self.conns:
                    if c.poll(0.5):
                        msg = c.recv()
                        raise RuntimeError('aggregation worker %d failed:\n%s' % (msg[1], msg[2]))
                raise RuntimeError('an aggregation worker failed')
            if time.time() - t0 > timeout:
                raise TimeoutError('aggregation workers %s did not finish within %.0f s' % (
                    [int(w) for w in np.nonzero(done != self.seq)[0]], timeout))
            time.sleep(0.00005)
        self.source_s = float(self.secs[_CTL_WORDS + 2 * W:_CTL_WORDS + 3 * W].max())
        self.aggregate_s = float(self.secs[_CTL_WORDS + 3 * W:_CTL_WORDS + 4 * W].max())
        return slot

    def close(self):
        if getattr(self, 'ctl', None) is not None:
            self.ctl[3] = 1
        for p in self.procs:
            p.join(5.0)
            if p.is_alive():
                p.terminate()
        for c in self.conns:
            c.close()
        self.conns, self.procs = [], []
        self.ctl = self.secs = None
        if getattr(self, 'ctl_blk', None) is not None:
            try:
                self.ctl_blk.close()
                self.ctl_blk.unlink()
            except Exception:
                pass
            self.ctl_blk = None


class PooledDataPrefetcher(LearnerDataPrefetcher):
    """LearnerDataPrefetcher whose aggregation runs in `pool`'s worker processes: the prefetch thread only hands out
    row ranges and submits the filled slot for its host-to-device copy"""

    def __init__(self, session_config, batch_size, pool):
        super().__init__(session_config, batch_size, stager=pool.stager)
        self.pool = pool

    def reset_stage_times(self):
        self.stage_s = {'wait_slot': 0.0, 'fill': 0.0, 'submit': 0.0, 'hand_over': 0.0, 'batches': 0}

    def run(self):
        try:
            if self.stager.on_gpu:
                torch.cuda.set_device(self.stager.device)
            self.reset_stage_times()
            while not self._stop.is_set():
                t0 = time.time()
                slot, _ = self.stager.begin_fill()
                t1 = time.time()
                self.pool.fill(slot, self.batch_size)
                t2 = time.time()
                data = ('slot', self.stager.submit(slot))
                t3 = time.time()
                while not self._stop.is_set():
                    try:
                        self.preprocess_queue.put(data, timeout=0.05)
                        break
                    except queue.Full:
                        pass
                st = self.stage_s
                st['wait_slot'] += t1 - t0; st['fill'] += t2 - t1; st['submit'] += t3 - t2; st['hand_over'] += time.time() - t3
                st['batches'] += 1
        except Exception as e:
            self.error = e
