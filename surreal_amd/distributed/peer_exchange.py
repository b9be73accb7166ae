"""
PeerExchange -- the learner ranks' collectives over IPC-mapped peer buffers (csrc/smx_xchg.hip,
include/surreal_amd.h "Data-parallel exchange").  One process per GPU; every rank allocates an
exchange buffer through the C ABI, the 64-byte IPC handles travel once through the
``torch.distributed`` group the learner already has, every rank maps its peers' buffers, and
from then on an all-reduce / all-gather is ONE kernel on the learner's stream -- capturable in
its hipGraph, no RCCL call, no graph cut.

What it replaces is ``torch.distributed.all_reduce`` / ``all_gather_into_tensor`` for the
learner's small latency-bound exchanges (per-epoch gradients + loss partial rows,
advantage moments, end-of-learn statistics).  The reference has one learner and no counterpart.

It is an optimisation with a verified fallback, never a requirement: ``PeerExchange.create``
returns None (and says why) when the ranks do not share a node, IPC is refused, or the
self-check -- random vectors against the process group's own all-reduce, plus bit-equality of
the result on all ranks -- fails; the learner then keeps using the process group.
"""
import ctypes
import logging

import torch

from surreal_amd import _lib as L

log = logging.getLogger('surreal_amd.peer_exchange')


class PeerExchange(object):
    last_failure = None        # why the most recent create() returned None (reported by bench.py / the self-test)

    def __init__(self, dist, capacity, timeout_s=2.0):
        """collective: every rank of `dist`'s default group constructs it with the same capacity
        (floats the largest exchange carries).  Raises SmxError / RuntimeError on failure."""
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if not 2 <= self.world <= 8:
            raise RuntimeError('PeerExchange serves 2..8 ranks of one node, got %d' % self.world)
        self.capacity = int(capacity)
        self.lib = L.load()
        self.device = torch.device('cuda', torch.cuda.current_device())
        nbytes = self.lib.smx_xchg_bytes(self.capacity, self.world)
        # Set-up is collective, and so are its failures: a rank whose allocation / export / mapping fails still takes part
        # in the handle gather and in the final agreement, so that EVERY rank raises (and PeerExchange.create falls back
        # on every rank) instead of one rank leaving its peers inside a collective.
        backend = dist.get_backend()
        where = self.device if backend == 'nccl' else 'cpu'
        self._own, self._opened, self.memory_kind = None, [], '?'
        problem = None
        handle = (ctypes.c_uint8 * 64)()
        try:
            own, kind = ctypes.c_void_p(), ctypes.c_int32()
            L.call('smx_xchg_alloc', nbytes, float(timeout_s), ctypes.byref(own), ctypes.byref(kind), L.current_stream())
            self._own, self.memory_kind = own, ('uncached', 'fine-grained', 'device')[kind.value]
            L.call('smx_xchg_export', own, handle)
        except Exception as e:
            problem = 'allocation / export: %r' % (e,)
        # the handles through the process group (64 bytes + an ok byte per rank; a CPU tensor for gloo, a device one for RCCL)
        mine = torch.tensor(list(bytes(handle)) + [0 if problem else 1], dtype=torch.uint8, device=where)
        every = torch.empty(self.world * 65, dtype=torch.uint8, device=where)
        dist.all_gather_into_tensor(every, mine)
        every = every.cpu().numpy().reshape(self.world, 65)
        if problem is None and not every[:, 64].all():
            problem = 'rank(s) %s could not allocate / export' % [int(r) for r in range(self.world) if not every[r, 64]]
        self.x = L.Xchg()
        self.x.world, self.x.rank, self.x.capacity = self.world, self.rank, self.capacity
        if problem is None:
            try:
                for p in range(self.world):
                    if p == self.rank:
                        self.x.peer[p] = self._own.value
                        continue
                    h = (ctypes.c_uint8 * 64)(*every[p, :64].tolist())
                    mapped = ctypes.c_void_p()
                    L.call('smx_xchg_open', h, ctypes.byref(mapped))
                    self._opened.append(mapped)
                    self.x.peer[p] = mapped.value
            except Exception as e:
                problem = 'mapping a peer buffer: %r' % (e,)
        self._status = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.exchanges = 0
        torch.cuda.synchronize()
        ok = torch.tensor([0.0 if problem else 1.0], device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)        # (also the barrier: nobody polls a buffer that is not mapped yet)
        if float(ok) < 1.0:
            self._release()
            raise RuntimeError('peer exchange set-up failed: %s' % (problem or 'on another rank'))

    def _release(self):
        for m in self._opened:
            self.lib.smx_xchg_close(m)
        self._opened = []
        if self._own is not None:
            self.lib.smx_xchg_free(self._own)
            self._own = None

    # ---- the two collectives ---------------------------------------------------------------------
    def all_reduce(self, t, err=None):
        """in place: t <- sum over ranks (identical bits on every rank)"""
        assert t.is_contiguous() and t.dtype == torch.float32 and t.numel() <= self.capacity
        L.call('smx_xchg_allreduce_f32', ctypes.byref(self.x), L.ptr(t), L.ptr(t), t.numel(), L.ptr(err),
               L.current_stream())
        self.exchanges += 1

    def all_gather_into_tensor(self, out, t, err=None):
        """out [world * n] <- every rank's t [n]"""
        assert t.is_contiguous() and out.is_contiguous() and t.dtype == out.dtype == torch.float32
        assert out.numel() == self.world * t.numel() and out.numel() <= self.capacity
        L.call('smx_xchg_allgather_f32', ctypes.byref(self.x), L.ptr(t), t.numel(), L.ptr(out), L.ptr(err),
               L.current_stream())
        self.exchanges += 1

    def status(self):
        """(exchanges completed on the device, error word) -- synchronises; diagnostics only"""
        L.call('smx_xchg_status', ctypes.byref(self.x), L.ptr(self._status), L.current_stream())
        s = self._status.cpu()
        return int(s[0]) & 0xffffffff, int(s[1]) & 0xffffffff

    # ---- validation before use ---------------------------------------------------------------------
    def self_check(self, rounds=16):
        """`rounds` all-reduces and all-gathers of changing vectors against the process group's results,
        plus bit-equality across ranks.  Collective; returns (ok, message).  Every rank runs EVERY round whatever it
        has seen so far: a rank that left early would issue a different sequence of process-group collectives than its
        peers (after a device error the exchange's own waits return at once, so a broken exchange costs no time)."""
        g = torch.Generator(device='cpu').manual_seed(1234 + self.rank)
        sizes = [self.capacity, max(4, self.capacity // 3 + 1), 3, 1031][:max(2, min(4, rounds))]
        fail = None
        for r in range(rounds):
            n = min(self.capacity, sizes[r % len(sizes)])
            a = (torch.rand(n, generator=g) - 0.5).to(self.device)
            want = a.clone()
            self.dist.all_reduce(want)
            got = a.clone()
            try:                  # a launch failure on THIS rank must not take it out of the rounds' process-group calls
                self.all_reduce(got)
            except Exception as e:
                fail = fail or 'all-reduce round %d raised %r' % (r, e)
            if fail is None and not torch.allclose(got, want, rtol=1e-5, atol=1e-6):
                fail = 'all-reduce round %d (n = %d): max diff %g' % (r, n, float((got - want).abs().max()))
            # replicas must be bit-identical: compare a checksum of the raw bits through the group
            bits = got.view(torch.int32).to(torch.int64)
            ck = torch.stack([bits.sum(), (bits * torch.arange(1, n + 1, device=self.device) % 1000003).sum()]).double()
            lo, hi = ck.clone(), ck.clone()
            self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
            self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
            if fail is None and not torch.equal(lo, hi):
                fail = 'all-reduce round %d: ranks hold different bits' % r
            m = min(n, self.capacity // self.world)
            part = a[:m].contiguous()
            want_g = torch.empty(self.world * m, device=self.device)
            self.dist.all_gather_into_tensor(want_g, part)
            got_g = torch.empty(self.world * m, device=self.device)
            try:
                self.all_gather_into_tensor(got_g, part)
            except Exception as e:
                fail = fail or 'all-gather round %d raised %r' % (r, e)
            if fail is None and not torch.equal(got_g, want_g):
                fail = 'all-gather round %d (n = %d) differs' % (r, m)
        done, err = self.status()
        if err:
            return False, 'device error word 0x%x (timeout | phase << 4 | peer)%s' % (err, '; ' + fail if fail else '')
        if fail is not None:
            return False, fail
        return True, '%d rounds, memory %s' % (rounds, self.memory_kind)

    @classmethod
    def create(cls, dist, capacity, timeout_s=2.0, rounds=16):
        """collective constructor with the fallback decision made IDENTICALLY on every rank: returns a
        checked PeerExchange, or None when any rank could not set it up or the self-check failed anywhere"""
        ex, why = None, ''
        try:
            ex = cls(dist, capacity, timeout_s)
        except Exception as e:               # IPC refused, allocation failed, ...
            why = repr(e)
        ok = torch.tensor([1.0 if ex is not None else 0.0])
        if dist.get_backend() == 'nccl':
            ok = ok.cuda()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) < 1.0:
            if ex is not None:
                ex.close()
            log.warning('peer exchange unavailable on some rank (%s): using the process group', why or 'a peer failed')
            cls.last_failure = 'set-up: %s' % (why or 'failed on another rank')
            return None
        try:
            good, msg = ex.self_check(rounds)
        except Exception as e:               # a launch failure / SmxError on THIS rank: still take part in the agreement
            good, msg = False, 'self-check raised %r' % (e,)
        ok = torch.tensor([1.0 if good else 0.0])
        if dist.get_backend() == 'nccl':
            ok = ok.cuda()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) < 1.0:
            log.warning('peer exchange self-check failed (%s): using the process group', msg)
            cls.last_failure = 'self-check: %s' % (msg if not good else 'failed on another rank')
            ex.close()
            return None
        ex.check_message = msg
        cls.last_failure = None
        return ex

    def close(self):
        torch.cuda.synchronize()
        self._release()
