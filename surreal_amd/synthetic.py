"""
Seeded synthetic PPO/DDPG workloads (numpy ``RandomState`` -- a frozen bit
stream, identical on every box) for the configurations BASELINE.json names.

There is no reference counterpart: the reference only ever sees real simulator
data.  Shapes follow the batch contract of ``MultistepAggregatorWithInfo``
(reference surreal/learner/aggregator.py:178-184):

    obs['low_dim']['flat_inputs'] (B, N, D)   obs_next[...] (B, 1, D)
    actions (B, N, A)  rewards (B, N)  dones (B, N) float32
    persistent_infos [ (B, N, 2A) ]  (behaviour policy [mean | std])
    onetime_infos None | [ h (B, L, hid), c (B, L, hid) ]

``on_policy=True`` draws the behaviour mean near zero and samples the actions
from it so that likelihoods sit above the reference's 1e-5 clamp
(ppo_net.py:46) and the surrogate has a non-zero gradient; ``False`` is the
SURVEY.md section 8(d) recipe (behaviour mean = tanh(N(0,1)), actions =
clip(N(0,1))), which lands every likelihood on the clamp -- the degenerate
branch the reference also has to survive.
"""
import collections

import numpy as np

PPO_CONFIGS = {
    # name: B, N, D, A  
    'cfg1_unit': dict(B=2, N=25, D=17, A=6),
    'cfg2_cheetah64': dict(B=64, N=128, D=17, A=6),
    'cfg5_synth1024': dict(B=1024, N=128, D=376, A=17),
    'tiny': dict(B=8, N=12, D=11, A=3),
    'ragged': dict(B=37, N=19, D=29, A=5),
}


def make_ppo_batch(B, N, D, A, seed=0, done_prob=0.01, on_policy=True,
                   init_log_sig=-1.0, rnn_hidden=0, rnn_layers=1, pixel=None):
    """pixel = (C, H, W) adds uint8 camera frames obs['pixel']['camera0'] (B, N, C, H, W) and
    obs_next (B, 1, C, H, W), drawn after everything else (older cases keep their bits)"""
    rs = np.random.RandomState(seed)
    obs = rs.randn(B, N, D).astype(np.float32)
    obs_next = rs.randn(B, 1, D).astype(np.float32)
    rewards = rs.randn(B, N).astype(np.float32)
    dones = (rs.rand(B, N) < done_prob).astype(np.float32)
    std = np.full((B, N, A), np.exp(init_log_sig), dtype=np.float32)
    if on_policy:
        mean = (0.05 * rs.randn(B, N, A)).astype(np.float32)
        actions = mean + std * rs.randn(B, N, A).astype(np.float32)
        actions = np.clip(actions, -1.0, 1.0).astype(np.float32)
    else:
        mean = np.tanh(rs.randn(B, N, A)).astype(np.float32)
        actions = np.clip(rs.randn(B, N, A), -1.0, 1.0).astype(np.float32)
    pds = np.concatenate([mean, std], axis=-1).astype(np.float32)
    onetime = None
    if rnn_hidden:
        onetime = [(0.1 * rs.randn(B, rnn_layers, rnn_hidden)).astype(np.float32),
                   (0.1 * rs.randn(B, rnn_layers, rnn_hidden)).astype(np.float32)]
    obs_d = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=obs))
    obs_next_d = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=obs_next))
    if pixel is not None:
        C, H, W = pixel
        obs_d['pixel'] = collections.OrderedDict(
            camera0=rs.randint(0, 256, (B, N, C, H, W)).astype(np.uint8))
        obs_next_d['pixel'] = collections.OrderedDict(
            camera0=rs.randint(0, 256, (B, 1, C, H, W)).astype(np.uint8))
    return {
        'obs': obs_d,
        'obs_next': obs_next_d,
        'actions': actions,
        'rewards': rewards,
        'dones': dones,
        'persistent_infos': [pds],
        'onetime_infos': onetime,
    }


def make_ppo_params(D, A, hidden=(300, 200), seed=1, init_log_sig=-1.0,
                    rnn_hidden=0, final_scale=0.05, pixel=None, cnn_feature_dim=256, rnn_layers=1,
                    log_sig_spread=0.0):
    """
    Canonical flat parameter dict (numpy fp32), the *injected* initial state
    for both the oracle and the HIP path.  Kaiming-uniform-like fan-in scaling
    (what torch.nn.Linear would draw) but from RandomState so that the GPU box
    regenerates the same bits; the last actor layer is scaled down so the
    initial policy mean is near zero.
    """
    rs = np.random.RandomState(seed)
    h1, h2 = hidden
    p = collections.OrderedDict()

    def lin(name, out_f, in_f, scale=1.0):
        bound = 1.0 / np.sqrt(in_f)
        p[name + '.W'] = (scale * rs.uniform(-bound, bound, (out_f, in_f))).astype(np.float32)
        p[name + '.b'] = (scale * rs.uniform(-bound, bound, (out_f,))).astype(np.float32)

    Dx = D                                   # width after the optional CNN stem (ppo_net.py:145,155)
    if pixel is not None:
        # CNNStemNetwork (builders.py:8-21), drawn from its own stream (older cases keep their bits)
        rc = np.random.RandomState(seed + 1000)
        C, H, W = pixel
        h1_, w1_ = (H - 8) // 4 + 1, (W - 8) // 4 + 1
        h2_, w2_ = (h1_ - 4) // 2 + 1, (w1_ - 4) // 2 + 1

        def cnn(name, shape):
            bound = 1.0 / np.sqrt(np.prod(shape[1:]))
            p['cnn.%s.W' % name] = rc.uniform(-bound, bound, shape).astype(np.float32)
            p['cnn.%s.b' % name] = rc.uniform(-bound, bound, shape[:1]).astype(np.float32)
        cnn('conv1', (16, C, 8, 8))
        cnn('conv2', (32, 16, 4, 4))
        cnn('fc', (cnn_feature_dim, 32 * h2_ * w2_))
        Dx = D + cnn_feature_dim
    in_f = rnn_hidden if rnn_hidden else Dx
    if rnn_hidden:
        bound = 1.0 / np.sqrt(rnn_hidden)
        p['rnn.weight_ih'] = rs.uniform(-bound, bound, (4 * rnn_hidden, Dx)).astype(np.float32)
        p['rnn.weight_hh'] = rs.uniform(-bound, bound, (4 * rnn_hidden, rnn_hidden)).astype(np.float32)
        p['rnn.bias_ih'] = rs.uniform(-bound, bound, (4 * rnn_hidden,)).astype(np.float32)
        p['rnn.bias_hh'] = rs.uniform(-bound, bound, (4 * rnn_hidden,)).astype(np.float32)
        if rnn_layers > 1:               # stacked layers (nn.LSTM names them *_l1, *_l2, ...); drawn
            rl = np.random.RandomState(seed + 2000)          # from their own stream: older cases keep their bits
            for layer in range(1, rnn_layers):
                sfx = '_l%d' % layer
                p['rnn.weight_ih' + sfx] = rl.uniform(-bound, bound, (4 * rnn_hidden, rnn_hidden)).astype(np.float32)
                p['rnn.weight_hh' + sfx] = rl.uniform(-bound, bound, (4 * rnn_hidden, rnn_hidden)).astype(np.float32)
                p['rnn.bias_ih' + sfx] = rl.uniform(-bound, bound, (4 * rnn_hidden,)).astype(np.float32)
                p['rnn.bias_hh' + sfx] = rl.uniform(-bound, bound, (4 * rnn_hidden,)).astype(np.float32)
    lin('actor.fc1', h1, in_f)
    lin('actor.fc2', h2, h1)
    lin('actor.fc3', A, h2, scale=final_scale)
    p['actor.log_var'] = np.full((1, A), init_log_sig, dtype=np.float32)
    if log_sig_spread:                   # a different log-sigma per action dimension (agent fixtures)
        p['actor.log_var'] += (log_sig_spread * np.linspace(-1.0, 1.0, A)).astype(np.float32)[None]
    lin('critic.fc1', h1, in_f)
    lin('critic.fc2', h2, h1)
    lin('critic.fc3', 1, h2)
    return p


def make_zfilter_state(D, seed=2, prewarm_rows=4096, eps=1e-5):
    """z-filter buffers after one z_update on N(0.3, 1.5) rows (reference init
    z_filter.py:40-42 then z_update :55-57)"""
    rs = np.random.RandomState(seed)
    x = (0.3 + 1.5 * rs.randn(prewarm_rows, D)).astype(np.float32)
    s = np.zeros(D, np.float32) + x.sum(0, dtype=np.float32)
    sq = np.full(D, eps, np.float32) + (x * x).sum(0, dtype=np.float32)
    cnt = np.array([eps], np.float32) + np.float32(prewarm_rows)
    return {'running_sum': s.astype(np.float32), 'running_sumsq': sq.astype(np.float32),
            'count': cnt.astype(np.float32)}


def make_ddpg_batch(B=512, D=17, A=6, seed=0, pixel=None):
    """SSARAggregator contract (aggregator.py:97-103); pixel = (C, H, W) adds
    uint8 camera frames obs['pixel']['camera0'] (B, C, H, W), drawn after everything else"""
    rs = np.random.RandomState(seed)
    out = _ddpg_low_dim(rs, B, D, A)
    if pixel is not None:
        for key in ('obs', 'obs_next'):
            out[key]['pixel'] = collections.OrderedDict(
                camera0=rs.randint(0, 256, size=(B,) + tuple(pixel)).astype(np.uint8))
    return out


def _ddpg_low_dim(rs, B, D, A):
    return {
        'obs': collections.OrderedDict(low_dim=collections.OrderedDict(
            flat_inputs=rs.randn(B, D).astype(np.float32))),
        'obs_next': collections.OrderedDict(low_dim=collections.OrderedDict(
            flat_inputs=rs.randn(B, D).astype(np.float32))),
        'actions': np.clip(rs.randn(B, A), -1, 1).astype(np.float32),
        'rewards': rs.randn(B, 1).astype(np.float32),
        'dones': (rs.rand(B, 1) < 0.02).astype(np.float32),
    }


def ppo_experiences(batch, lo=0, hi=None):
    """rows [lo, hi) of a make_ppo_batch() batch as the per-step Python objects a collector hands over (SURVEY.md
    Appendix B.1: lists of per-step observation dicts / action arrays / float rewards / bool dones / [pd] infos)"""
    ob, obn = batch['obs']['low_dim']['flat_inputs'], batch['obs_next']['low_dim']['flat_inputs']
    pi = batch['persistent_infos'][0]
    N = ob.shape[1]
    hi = ob.shape[0] if hi is None else hi
    return [{'obs': [{'low_dim': {'flat_inputs': ob[b, s]}} for s in range(N)],
             'obs_next': {'low_dim': {'flat_inputs': obn[b, 0]}},
             'actions': [batch['actions'][b, s] for s in range(N)],
             'rewards': [float(x) for x in batch['rewards'][b]], 'dones': [bool(x) for x in batch['dones'][b]],
             'persistent_infos': [[pi[b, s]] for s in range(N)], 'onetime_infos': [], 'n_step': N}
            for b in range(lo, hi)]


class SyntheticExperienceSource(object):
    """picklable source_factory for surreal_amd.distributed.AggregationPool: worker w serves source(n, seq, row_lo) ->
    rows [row_lo, row_lo + n) of batch `seq % distinct` (seed0 + that), as per-step Python objects.  fresh=True rebuilds
    the objects on every call from a pickled copy -- what a worker that has just deserialised a chunk from the collector
    holds -- instead of handing out the same objects again."""

    def __init__(self, B, N, D, A, seed0=100, distinct=1, fresh=False):
        self.args = (B, N, D, A, seed0, distinct, fresh)

    def __call__(self, worker_index):
        import pickle
        B, N, D, A, seed0, distinct, fresh = self.args
        cache = {}

        def source(n, seq, lo):
            key = (seq % distinct, lo, n)
            if key not in cache:
                batch = make_ppo_batch(B, N, D, A, seed=seed0 + key[0])
                exps = ppo_experiences(batch, lo, lo + n)
                cache[key] = pickle.dumps(exps, protocol=4) if fresh else exps
            return pickle.loads(cache[key]) if fresh else cache[key]
        return source
