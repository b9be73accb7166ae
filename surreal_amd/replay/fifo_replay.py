"""
FIFOReplay (surreal/replay/fifo_replay.py:6-49): ``deque(maxlen=memory_size + 3)``,
``insert = append`` (silently dropping the oldest on overflow), ``sample = popleft x batch``,
ready when ``len >= batch_size``.  On-policy PPO uses it as a conveyor belt between rollout
workers and the learner.
"""
from collections import deque

import torch

from .base import Replay


class FIFOReplay(Replay):
    def __init__(self, learner_config, env_config, session_config, index=0):
        super().__init__(learner_config=learner_config, env_config=env_config,
                         session_config=session_config, index=index)
        self.batch_size = self.learner_config.replay.batch_size
        self.memory_size = self.learner_config.replay.memory_size
        self._memory = deque(maxlen=self.memory_size + 3)   # "+ 3 for a gentle buffering"
        # the reference asserts the transport is configured for on-policy use (:28-31)
        assert self.session_config.replay.max_puller_queue <= 10
        assert self.session_config.replay.get('max_prefetch_queue', 1) == 1
        assert not self.session_config.sender.get('flush_time', 0)
        assert self.session_config.sender.flush_iteration <= 10
        # device tier: a ring of capacity memory_size + 3 with a head and a count
        self._head = 0
        self._count = 0

    # ---- host tier: the reference semantics ---------------------------------------------
    def insert(self, exp_tuple):
        self._memory.append(exp_tuple)

    def sample(self, batch_size):
        assert batch_size <= self.memory_size
        return [self._memory.popleft() for _ in range(batch_size)]

    def evict(self):
        raise NotImplementedError('no support for eviction in FIFO mode')

    def start_sample_condition(self):
        return len(self) >= self.batch_size

    def __len__(self):
        return len(self._memory) + self._count

    # ---- device tier: same FIFO semantics over HBM tables --------------------------------
    def insert_batch(self, fields):
        """fields: {name: device tensor [n, ...]}; appends n experiences (oldest dropped when
        more than memory_size + 3 are held, like deque(maxlen=...))"""
        cap = self.memory_size + 3
        tables = self._ensure_tables(cap, fields)
        n = next(iter(fields.values())).shape[0]
        if n > cap:                                   # only the newest `cap` survive an append storm
            fields = {k: v[n - cap:] for k, v in fields.items()}
            n = cap
        tail = (self._head + self._count) % cap
        for name, t in fields.items():
            tables[name].insert(tail, t)
        overflow = max(0, self._count + n - cap)
        self._head = (self._head + overflow) % cap
        self._count = min(cap, self._count + n)
        self.cumulative_collected_count += n

    def reserve_batch(self, n, shapes, dtypes=None):
        """Zero-copy insert, step 1: views of the next `n` table rows per field, for a producer that
        writes experiences in place (SyntheticVecEnv.emit_windows(out=...)); None when the rows
        would wrap around the ring or overflow it (fall back to insert_batch).
        shapes: {name: per-experience shape}; dtypes: {name: torch dtype} for the fields that are not fp32 (uint8
        frames).  Follow with commit_batch(n)."""
        cap = self.memory_size + 3
        dtypes = dtypes or {}
        fields = {k: torch.empty((0,) + tuple(shp), dtype=dtypes.get(k, torch.float32)) for k, shp in shapes.items()}
        tables = self._ensure_tables(cap, fields)
        tail = (self._head + self._count) % cap
        if tail + n > cap and self._count == 0 and n <= cap:
            # an EMPTY ring whose next rows would wrap: start over at row 0 (nothing is stored, so nothing moves; views
            # popped earlier than the most recent pop must no longer be in use -- a learner consumes a pop before the
            # next one).  A producer / consumer pair that moves whole batches then alternates between two fixed row
            # ranges: two address sets, two captured graphs, no staging copy.
            self._head = tail = 0
        if n > cap - self._count or tail + n > cap:
            return None
        return {name: tables[name].rows(tail, n) for name in shapes}

    def commit_batch(self, n):
        """zero-copy insert, step 2: the reserved rows now hold experiences"""
        assert n <= self.memory_size + 3 - self._count
        self._count += n
        self.cumulative_collected_count += n

    def sample_batch(self, batch_size, copy=True):
        """pops the `batch_size` oldest device-tier experiences -> {name: [batch, ...]}.
        copy=False returns VIEWS of the table when the popped rows are contiguous: valid until the
        ring wraps over them, i.e. for a consumer that is enqueued before the next inserts (the
        learner, which stages a moving batch itself)."""
        assert batch_size <= self.memory_size and batch_size <= self._count
        cap = self.memory_size + 3
        if not copy and self._head + batch_size <= cap:
            out = {name: tab.rows(self._head, batch_size) for name, tab in self._tables.items()}
            self._head = (self._head + batch_size) % cap
            self._count -= batch_size
            self.cumulative_sampled_count += batch_size
            return out
        idx = (self._head + torch.arange(batch_size, device=self._dev)) % cap
        out = {name: tab.gather(idx) for name, tab in self._tables.items()}
        self._head = (self._head + batch_size) % cap
        self._count -= batch_size
        self.cumulative_sampled_count += batch_size
        return out
