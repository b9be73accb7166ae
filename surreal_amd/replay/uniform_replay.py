"""
UniformReplay (surreal/replay/uniform_replay.py:6-74): a ring buffer with an insert cursor
``_next_idx``; ``sample`` draws ``batch_size`` indices uniformly WITH replacement
(``random.randint(0, len - 1)`` each); ready when ``len > sampling_start_size``.
"""
import random

import torch

from .base import Replay


class UniformReplay(Replay):
    def __init__(self, learner_config, env_config, session_config, index=0):
        super().__init__(learner_config=learner_config, env_config=env_config,
                         session_config=session_config, index=index)
        self._memory = []
        self.memory_size = self.learner_config.replay.memory_size
        self._next_idx = 0
        # device tier
        self._dev_len = 0
        self._dev_next = 0
        self._draws = 0
        self.seed = int(self.session_config.replay.get('seed', 0)) + 7919 * int(index)

    # ---- host tier: the reference semantics ---------------------------------------------
    def insert(self, exp_dict):
        if self._next_idx >= len(self._memory):
            self._memory.append(exp_dict)
        else:
            self._memory[self._next_idx] = exp_dict
        self._next_idx = (self._next_idx + 1) % self.memory_size

    def sample(self, batch_size):
        indices = [random.randint(0, len(self._memory) - 1) for _ in range(batch_size)]
        return [self._memory[i] for i in indices]

    def evict(self):
        raise NotImplementedError

    def start_sample_condition(self):
        return len(self) > self.learner_config.replay.sampling_start_size

    def __len__(self):
        return len(self._memory) + self._dev_len

    # ---- device tier -----------------------------------------------------------------------
    def insert_batch(self, fields):
        tables = self._ensure_tables(self.memory_size, fields)
        n = next(iter(fields.values())).shape[0]
        if n > self.memory_size:
            fields = {k: v[n - self.memory_size:] for k, v in fields.items()}
            n = self.memory_size
        for name, t in fields.items():
            tables[name].insert(self._dev_next, t)
        self._dev_next = (self._dev_next + n) % self.memory_size
        self._dev_len = min(self.memory_size, self._dev_len + n)
        self.cumulative_collected_count += n

    def sample_indices(self, batch_size):
        """with-replacement uniform indices on the device (Philox4x32-10; the reference's
        Python Mersenne stream cannot be reproduced on a GPU -- distributional parity, and
        sample_batch(indices=...) takes injected indices for exact-parity tests)"""
        idx = torch.empty(batch_size, dtype=torch.int64, device=self._dev)
        self._K.uniform_indices(idx, self._dev_len, self.seed, self._draws)
        self._draws += batch_size
        return idx

    def sample_batch(self, batch_size, indices=None, out=None):
        """one launch for the whole sample (smx_uniform_gather_multi): every field's rows, the indices drawn where they
        are used -- the same Philox counters as sample_indices(), so the two agree row for row.
        out: {field: tensor} to gather INTO (contiguous, the field's dtype, batch_size rows) -- e.g. a learner's staging
        buffers (DDPGLearner.staging_fields): the batch then needs no copy on its way into the captured iteration"""
        idx = None if indices is None else torch.as_tensor(indices, dtype=torch.int64).to(self._dev)
        self.cumulative_sampled_count += batch_size
        names = list(self._tables)
        if len(names) > 8:                       # (more fields than one launch carries: field by field)
            idx = self.sample_indices(batch_size) if idx is None else idx
            return {name: tab.gather(idx) for name, tab in self._tables.items()}
        tabs = [self._tables[k] for k in names]
        outs = []
        for k, t in zip(names, tabs):
            o = None if out is None else out.get(k)
            if o is None:
                o = torch.empty(batch_size, t.width, device=self._dev, dtype=t.dtype)
            else:
                assert (o.is_contiguous() and o.dtype == t.dtype and o.numel() == batch_size * t.width
                        and o.shape[0] == batch_size), k
                o = o.view(batch_size, t.width)
            outs.append(o)
        self._K.uniform_gather_multi([t.data for t in tabs], outs, self._dev_len, self.seed, self._draws, idx=idx)
        if idx is None:
            self._draws += batch_size
        return {k: o.view((batch_size,) + t.shape) for k, o, t in zip(names, outs, tabs)}
