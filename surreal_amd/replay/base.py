"""
Replay plugin base class: the constructor signature, abstract methods and counters of the
reference's ``surreal.replay.base.Replay`` (surreal/replay/base.py:9-256) without its ZeroMQ
servers.  The reference runs a collector thread (agents PUSH pyarrow blobs) and a sampler
thread (learner REQ -> ``sample`` -> pyarrow blob); here agents and the learner live in the same
process (one per GPU) and call ``insert`` / ``sample`` directly, so ``start_threads`` / ``join``
are no-ops kept for call-site compatibility.

Two storage tiers behind the same object:
  * host tier  -- ``insert(exp_dict)`` / ``sample(batch_size) -> list[exp]``: exactly the
    reference's Python semantics (what external, CPU-simulator agents use);
  * device tier -- ``insert_batch(fields)`` / ``sample_batch(batch_size)``: experience fields as
    struct-of-arrays tables in HBM moved by the ring-insert / row-gather HIP kernels
    (csrc/smx_replay.hip); what GPU-resident rollouts use, and what the benchmark path feeds
    the learner from without ever touching the host.
"""
import time

import torch

from surreal_amd import kernels as KN


def table_dtype(dtype):
    """element type a field is stored with: uint8 (camera frames, `pixel_input`) stays uint8 -- a quarter of the
    HBM and copy bytes of an fp32 widening, and what the CNN stem's im2col reads anyway; every other field
    (observations, actions, rewards, dones, policy infos) is fp32 like the learner's batch contract"""
    return torch.uint8 if dtype == torch.uint8 else torch.float32


class DeviceTable(object):
    """one experience field: [capacity, width] rows in HBM, fp32 or uint8 (``table_dtype``)"""

    def __init__(self, capacity, shape, device, kernels, dtype=torch.float32):
        self.shape = tuple(shape)
        width = 1
        for s in self.shape:
            width *= int(s)
        self.width = width
        self.dtype = table_dtype(dtype)
        self.data = torch.zeros(capacity, width, device=device, dtype=self.dtype)
        self.K = kernels

    def insert(self, cursor, rows):
        self.K.ring_insert(self.data, cursor, rows.reshape(rows.shape[0], -1).to(self.dtype).contiguous())

    def gather(self, idx):
        out = torch.empty(idx.numel(), self.width, device=self.data.device, dtype=self.dtype)
        self.K.gather_rows(self.data, idx, out)
        return out.view((idx.numel(),) + self.shape)

    def rows(self, first, n):
        """the n consecutive rows from `first` as a VIEW of the table, in the field's shape"""
        return self.data[first:first + n].view((n,) + self.shape)


class Replay(object):
    def __init__(self, learner_config, env_config, session_config, index=0):
        self.learner_config = learner_config
        self.env_config = env_config
        self.session_config = session_config
        self.index = index
        self._evict_interval = session_config.replay.get('evict_interval', 0.0)
        self._setup_logging()
        # device tier (created lazily by insert_batch)
        self._tables = None
        self._dev_capacity = 0

    # ---- lifecycle: in-process, nothing to start -----------------------------------------
    def start_threads(self):
        pass

    def join(self):
        pass

    # ---- abstract (base.py:69-113) --------------------------------------------------------
    def insert(self, exp_dict):
        raise NotImplementedError

    def sample(self, batch_size):
        raise NotImplementedError

    def evict(self):
        pass

    def start_sample_condition(self):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    # ---- counters / wrappers (base.py:115-171) ---------------------------------------------
    def _setup_logging(self):
        self.init_time = time.time()
        self.cumulative_collected_count = 0
        self.cumulative_sampled_count = 0
        self.cumulative_request_count = 0
        self.last_report_time = self.init_time

    def _insert_wrapper(self, exp):
        self.cumulative_collected_count += 1
        self.insert(exp)

    def _sample_request_handler(self, batch_size):
        """what the reference's sampler thread does per learner request (base.py:156-171),
        minus the 10 ms spin and the serialisation"""
        if not self.start_sample_condition():
            return None
        self.cumulative_sampled_count += batch_size
        self.cumulative_request_count += 1
        return self.sample(batch_size)

    def generate_tensorplex_report(self):
        now = time.time()
        dt = max(now - self.last_report_time, 1e-9)
        self.last_report_time = now
        return {
            'num_exps': len(self),
            'cumulative_exps': self.cumulative_collected_count,
            'cumulative_sampled': self.cumulative_sampled_count,
            'cumulative_requests': self.cumulative_request_count,
            'lifetime_experience_utilization_percent':
                self.cumulative_sampled_count / max(self.cumulative_collected_count, 1) * 100,
            'report_interval_s': dt,
        }

    # ---- device tier helpers ---------------------------------------------------------------
    def _ensure_tables(self, capacity, fields):
        if self._tables is None:
            K, dev = KN.default_kernels(), KN.default_device()
            self._tables = {name: DeviceTable(capacity, t.shape[1:], dev, K, t.dtype)
                            for name, t in fields.items()}
            self._dev_capacity = capacity
            self._K, self._dev = K, dev
        return self._tables
