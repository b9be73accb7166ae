"""
Experience aggregators: list of per-agent experience dicts -> batched arrays.

``MultistepAggregatorWithInfo`` produces exactly the batch contract of the reference class of
the same name (surreal/learner/aggregator.py:106-262, SURVEY.md Appendix B.1 -> B.2):

    obs[mod][key] (B, N, ...)    obs_next[mod][key] (B, 1, ...)   actions (B, N, A)
    rewards (B, N)   dones (B, N) float32   persistent_infos [ (B, N, 2A) ] | None
    onetime_infos [ (B, L, hid), (B, L, hid) ] | None

The reference builds it with nested Python loops over B x N tiny arrays (1.8 s for the
1024 x 128 batch); here each field is one ``np.asarray`` over the
already-contiguous per-experience lists: 0.16 s for the same batch in steady state (0.87 s the
first two calls, while the 197 MB result is touched for the first time; one ``np.concatenate`` over
the flat leaf list measures the same).  Experiences that already arrive as one array per
field (what the device-resident replay hands over) pass straight through ``np.stack``.
"""
import collections
import os

import numpy as np

_NATIVE = [False, None]          # [looked for, the extension's fill() or None]


def native_fill():
    """``_smx_host.fill`` (surreal_amd/csrc/host/smx_host.c, built by ``surreal_amd.build.build_host``) or None when the
    extension has not been built.  It writes one field of a batch into a preallocated array: leaf pointers collected with
    the GIL held, the copies done without it on a few threads.  SMX_HOST_NUMPY=1 forces the numpy path (A/B runs)."""
    if not _NATIVE[0]:
        _NATIVE[0] = True
        if not os.environ.get('SMX_HOST_NUMPY'):
            try:
                from surreal_amd import _smx_host
                _NATIVE[1] = _smx_host.fill
            except ImportError:
                import warnings
                warnings.warn('surreal_amd/_smx_host.so is not built (python -m surreal_amd.build): host-fed batches are '
                              'assembled by numpy, ~5x slower')
    return _NATIVE[1]


def _stack(seq, dtype=None, out=None):
    """`out`: a preallocated array of the result's shape (e.g. a pinned staging buffer: PinnedBatchStager.host_views)
    that receives the values in place -- the batch is then never materialised anywhere else on the host"""
    if out is not None:
        if len(seq) and isinstance(seq[0], (list, tuple)):          # [B][N] leaves -> one flat list
            flat = [leaf for row in seq for leaf in row]
            np.stack(flat, out=out.reshape((len(flat),) + out.shape[2:]), casting='unsafe')
        else:
            np.stack([np.asarray(x) for x in seq], out=out, casting='unsafe')
        return out
    a = np.asarray(seq)
    if a.dtype == object:      # ragged input is a contract violation, not something to paper over
        raise ValueError('experiences have inconsistent shapes')
    return a if dtype is None else a.astype(dtype, copy=False)


class FrameStackPreprocessor(object):
    """SSAR experiences whose camera frames were stacked as a LIST (``frame_stack_concatenate_on_env``
    off: the sender ships every frame once, exp_sender.py dedup) get the frames of each observation
    joined on the channel axis before aggregation (surreal/learner/aggregator.py:11-30, used by
    DDPGLearner._prefetcher_preprocess, ddpg.py:430-440).
    The reference rewrites the experiences it is handed, which is safe there only because every
    batch is a fresh copy off the replay socket.  Here the learner samples the replay's own objects
    (neighbouring SSAR experiences share one observation dict; uniform sampling repeats experiences),
    so the joined frames go into NEW experience / observation dicts and the sampled ones stay as the
    agent sent them."""

    def __init__(self, frame_stacks):
        self.frame_stacks = frame_stacks

    @staticmethod
    def preprocess_obs(obs):
        """-> a shallow copy of `obs` whose camera entries are single (C, H, W) arrays"""
        if 'pixel' not in obs:
            return obs
        out = type(obs)(obs)
        out['pixel'] = type(obs['pixel'])(obs['pixel'])
        for key, frames in obs['pixel'].items():
            joined = frames if isinstance(frames, np.ndarray) and frames.ndim == 3 \
                else np.concatenate(frames, axis=0)
            if joined.ndim != 3:
                raise AssertionError('stacked camera frames must join to (C, H, W)')
            out['pixel'][key] = joined
        return out

    def preprocess_list(self, exp_list):
        out = []
        for exp in exp_list:
            exp = dict(exp)
            exp['obs'] = [self.preprocess_obs(exp['obs'][0]), self.preprocess_obs(exp['obs'][1])]
            out.append(exp)
        return out


class MultistepAggregatorWithInfo(object):
    def __init__(self, obs_spec, action_spec):
        if not isinstance(obs_spec, dict) or not isinstance(action_spec, dict):
            raise TypeError('obs_spec and action_spec must be dicts')
        self.action_type = action_spec['type']
        self.action_spec = action_spec
        self.obs_spec = obs_spec
        # copy threads of the in-place path.  One by default: the leaf walk (GIL held) is half the time, and on hosts with
        # a CPU quota below their core count extra OpenMP threads spin instead of copying (measured 43 -> 105 ms at 8)
        self.threads = max(1, int(os.environ.get('SMX_HOST_THREADS', '1')))

    def _batch_obs(self, per_exp_steps, into=None, native=None):
        """per_exp_steps: list (B) of list (steps) of nested obs dicts (or a callable that builds it) -> dict of
        (B, steps, ...).  native = (exp_list, field, per_step): try the C extension first (in-place staging only)"""
        out = collections.OrderedDict()
        fill = native_fill() if (native is not None and into is not None) else None
        for modality in self.obs_spec.keys():
            out[modality] = collections.OrderedDict()
            for key in self.obs_spec[modality].keys():
                dst = None if into is None else into[modality][key]
                if fill is not None and fill(dst, native[0], native[1], (modality, key), native[2], self.threads) >= 0:
                    out[modality][key] = dst
                    continue
                if callable(per_exp_steps):          # (built only when some field needs the numpy path)
                    per_exp_steps = per_exp_steps()
                out[modality][key] = _stack(
                    [[step[modality][key] for step in steps] for steps in per_exp_steps], out=dst)
        return out

    def _field(self, exp_list, name, dst, path=(), per_step=1):
        """one plain field (actions / rewards / dones / an info slot) into `dst` through the extension; False: not taken"""
        fill = native_fill() if dst is not None else None
        return fill is not None and fill(dst, exp_list, name, path, per_step, self.threads) >= 0

    def _gather_action_infos(self, exp_list, into=None):
        """aggregator.py:223-262"""
        first = exp_list[0]
        onetime = persistent = None
        o = (lambda name, i: None) if into is None else (lambda name, i: into[name][i])
        if len(first['onetime_infos']) > 0:
            onetime = [o('onetime_infos', i) if self._field(exp_list, 'onetime_infos', o('onetime_infos', i), (i,), 0)
                       else _stack([exp['onetime_infos'][i] for exp in exp_list], out=o('onetime_infos', i))
                       for i in range(len(first['onetime_infos']))]
        if len(first['persistent_infos'][0]) > 0:
            persistent = [o('persistent_infos', i)
                          if self._field(exp_list, 'persistent_infos', o('persistent_infos', i), (i,), 1)
                          else _stack([[step[i] for step in exp['persistent_infos']] for exp in exp_list],
                                      out=o('persistent_infos', i))
                          for i in range(len(first['persistent_infos'][0]))]
        return onetime, persistent

    def aggregate(self, exp_list, out=None):
        """out: a batch-shaped tree of preallocated arrays (PinnedBatchStager.host_views(slot)) that receives every
        field in place -- experiences unpacked from the collector's chunks go straight into the pinned
        struct-of-arrays staging the host-to-device copy reads (no intermediate batch on the host)"""
        if self.action_type != 'continuous':
            # the reference's discrete branch is broken (aggregator.py:172-173) -- continuous only
            raise NotImplementedError('action_spec unsupported ' + str(self.action_spec))
        g = (lambda k: None) if out is None else (lambda k: out[k])
        observations = self._batch_obs(lambda: [exp['obs'] for exp in exp_list], g('obs'), (exp_list, 'obs', 1))
        next_obs = self._batch_obs(lambda: [[exp['obs_next']] for exp in exp_list], g('obs_next'),
                                   (exp_list, 'obs_next', 0))
        onetime, persistent = self._gather_action_infos(exp_list, out)

        def plain(name):
            dst = g(name)
            return dst if self._field(exp_list, name, dst) else _stack([exp[name] for exp in exp_list], out=dst)
        dones = plain('dones')
        return {
            'obs': observations,
            'obs_next': next_obs,
            'actions': plain('actions'),
            'rewards': plain('rewards'),
            'persistent_infos': persistent,
            'onetime_infos': onetime,
            'dones': dones if out is not None else dones.astype('float32'),
        }


class SSARAggregator(object):
    """(s, s', a, r, done) batches for DDPG (surreal/learner/aggregator.py:33-103,
    SURVEY.md Appendix B.3): obs / obs_next[mod][key] (B, ...), actions (B, A),
    rewards (B, 1), dones (B, 1)"""

    def __init__(self, obs_spec, action_spec):
        self.action_type = action_spec['type']
        self.action_spec = action_spec
        self.obs_spec = obs_spec

    def aggregate(self, exp_list):
        obs0 = collections.OrderedDict()
        obs1 = collections.OrderedDict()
        for modality in self.obs_spec.keys():
            obs0[modality] = collections.OrderedDict()
            obs1[modality] = collections.OrderedDict()
            for key in self.obs_spec[modality].keys():
                obs0[modality][key] = _stack([exp['obs'][0][modality][key] for exp in exp_list])
                obs1[modality][key] = _stack([exp['obs'][1][modality][key] for exp in exp_list])
        if self.action_type == 'continuous':
            actions = _stack([exp['action'] for exp in exp_list], np.float32)
        elif self.action_type == 'discrete':
            actions = _stack([exp['action'] for exp in exp_list], np.int32)
        else:
            raise NotImplementedError('action_spec unsupported ' + str(self.action_spec))
        rewards = _stack([exp['reward'] for exp in exp_list], np.float32)
        dones = _stack([float(exp['done']) for exp in exp_list], np.float32)
        return {
            'obs': obs0,
            'obs_next': obs1,
            'actions': actions,
            'rewards': np.expand_dims(rewards, axis=1),
            'dones': np.expand_dims(dones, axis=1),
        }
