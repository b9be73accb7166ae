"""
PPOLearner for MI355X -- drop-in for ``surreal.learner.ppo.PPOLearner``
(surreal/learner/ppo.py:12-682): same constructor ``(learner_config, env_config,
session_config)``, same config keys, same batch contract (``MultistepAggregatorWithInfo``),
same method names, same statistics keys, and (to 1e-5 fp32) the same advantages, returns,
per-epoch losses and updated parameters as the reference CPU learner on identical inputs.

What is different is how one ``learn()`` runs (SURVEY.md section 3.1 lists the reference flow):

  * trajectories stay in HBM as struct-of-arrays ``(B, N, .)`` tensors; the reference's
    ``torch.cat([obs, obs_next])`` copy is never made -- the fused critic kernel addresses
    both tensors directly;
  * the critic pass over all B*(N+1) steps is one fused z-filter + 3-layer MLP kernel on FP32
    MFMA (csrc/smx_mlp3_fused.hip); GAE, normalisation, losses, backward, clip-norm and Adam
    are HIP kernels chained on the stream with NO host synchronisation: the reference's ~15
    ``.item()`` reads per learn become one read-back of a small statistics block, and the
    data-dependent ``break`` of the policy loop (ppo.py:556-557) becomes a device-side flag that
    turns the remaining policy kernels into no-ops;
  * the forward pass that the reference repeats after every update just to measure KL
    (ppo.py:553) is the next epoch's forward pass, so it is computed once;
  * the value epochs (independent parameters, ppo.py:561-562) run on a second HIP stream next
    to the policy epochs, and the whole step is captured in a hipGraph and replayed;
  * with ``torch.distributed`` initialised (one process per GPU, RCCL over xGMI) each rank
    learns on its shard of the sub-trajectories and the ranks all-reduce the statistics the
    single reference learner would have seen whole: advantage moments, loss partial sums,
    gradients, value moments, z-filter sums.

Scope: MLP policy, LSTM-stem policy (``rnn.if_rnn_policy``, the reference default; stacked layers
with ``rnn.rnn_layer > 1``) and / or CNN stem over camera frames (``pixel_input``), one camera.
"""
import gc
import types

import numpy as np
import torch

from surreal_amd import _lib as L
from surreal_amd import kernels as KN
from surreal_amd.learner.aggregator import MultistepAggregatorWithInfo
from surreal_amd.learner.base import Learner, DeferredStats
from surreal_amd.model.ppo_net import DiagGauss, PPOModel


class LinearWithMinLR(object):
    """Learning-rate schedule named by ``algo.network.anneal.lr_scheduler`` (ppo.py:121-125,
    171-178).  The reference takes it from torchx 0.9 whose source is absent, so the exact
    formula is unpinned: this one anneals linearly from the initial rate to ``min_lr`` over
    ``num_updates`` calls of ``step()``, re-evaluated every ``update_freq`` calls."""

    def __init__(self, initial_lr, num_updates, update_freq=1, min_lr=0.0):
        self.initial_lr = float(initial_lr)
        self.num_updates = max(int(num_updates), 1)
        self.update_freq = max(int(update_freq), 1)
        self.min_lr = float(min_lr)
        self.n = 0
        self.lr = float(initial_lr)

    def get_lr(self):
        return [self.lr]

    def step(self):
        self.n += 1
        if self.n % self.update_freq == 0:
            frac = max(0.0, 1.0 - self.n / self.num_updates)
            self.lr = max(self.min_lr, self.initial_lr * frac)

    def state_dict(self):
        return {'n': self.n, 'lr': self.lr}

    def load_state_dict(self, sd):
        self.n, self.lr = int(sd['n']), float(sd['lr'])


class _SegmentedGraph(object):
    """A learn() on several ranks as hipGraph SEGMENTS with the collectives issued between them: the
    kernels (and the few torch ops) between two collectives are captured once and replayed, RCCL is
    called eagerly on the same stream.  One learn of the benchmark is ~70 launches; issued one by
    one from Python they cost more host time than the GPU needs to run them, which a single-rank
    learner never pays (its whole step is one graph)."""

    def __init__(self):
        self.items = []
        self.pool = torch.cuda.graph_pool_handle()      # one pool: a segment's temporaries outlive it
        self._cur = None

    def _open(self):
        g = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread queries events while we capture; under the
        # default (global) mode a HIP call from ANY thread invalidates the capture
        ctx = torch.cuda.graph(g, pool=self.pool, capture_error_mode='thread_local')
        ctx.__enter__()
        self._cur = (g, ctx)

    def _close(self):
        g, ctx = self._cur
        ctx.__exit__(None, None, None)
        self.items.append(g)
        self._cur = None

    def capture(self, fn, dist_proxy):
        dist_proxy.recorder = self
        self._open()
        try:
            fn()
        finally:
            self._close()
            dist_proxy.recorder = None

    def collective(self, thunk):
        """called by the distributed proxy while capturing: cut the graph here (the collective is NOT
        executed during the capture pass -- the kernels around it are not either)"""
        self._close()
        self.items.append(thunk)
        self._open()

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()


class _CountingDist(object):
    """torch.distributed with a counter on the collectives a learn() issues (reported by bench.py), a hook
    for _SegmentedGraph, and -- when the ranks share a node -- the fp32 exchanges routed through
    surreal_amd.distributed.PeerExchange (one kernel on the learner's stream, part of its graph) instead of
    the process group (an eager RCCL call that cuts the graph)."""

    def __init__(self, dist):
        self._d = dist
        self.count = 0
        self.recorder = None
        self.exchange = None         # PeerExchange, once a workspace has set it up and checked it
        self.err_word = None         # device int32 a timed-out exchange raises (the learner's control block)

    def _run(self, name, a, k):
        def thunk():
            self.count += 1
            return getattr(self._d, name)(*a, **k)
        if self.recorder is not None:
            self.recorder.collective(thunk)
            return None
        return thunk()

    def _peer_ok(self, *tensors):
        ex = self.exchange
        return ex is not None and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and
                                      t.numel() <= ex.capacity and t.data_ptr() % 16 == 0 for t in tensors)

    def all_reduce(self, t, *a, **k):
        if not a and not k and self._peer_ok(t):
            self.count += 1
            return self.exchange.all_reduce(t, err=self.err_word)
        return self._run('all_reduce', (t,) + a, k)

    def all_gather_into_tensor(self, out, t, *a, **k):
        if not a and not k and self._peer_ok(out, t):
            self.count += 1
            return self.exchange.all_gather_into_tensor(out, t, err=self.err_word)
        return self._run('all_gather_into_tensor', (out, t) + a, k)

    def __getattr__(self, name):
        return getattr(self._d, name)


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return _CountingDist(dist), dist.get_world_size(), dist.get_rank()
    return None, 1, 0


class PPOLearner(Learner):
    def __init__(self, learner_config, env_config, session_config):
        super().__init__(learner_config, env_config, session_config)
        self.K = KN.default_kernels()       # raises when the HIP library / GPU is missing
        self.device = KN.default_device()
        self._dist, self.world_size, self.rank = _dist_info()
        # statistics are read back asynchronously on a GPU (see _collect_stats)
        self._pending_stats, self._trace, self._epochs_executed, self._kl_record = None, None, 0, []
        self.lazy_stats = self.device != 'cpu' and bool(self.session_config.learner.get('lazy_stats', True))

        self.current_iteration = 0
        self.global_step = 0
        self.gpu_option = 'cuda:all'
        self.use_cuda = True

        algo = self.learner_config.algo
        # RL general parameters (ppo.py:75-85)
        self.gamma = algo.gamma
        self.lam = algo.advantage.lam
        self.n_step = algo.n_step
        self.use_z_filter = algo.use_z_filter
        self.use_r_filter = algo.use_r_filter
        self.norm_adv = algo.advantage.norm_adv
        self.batch_size = self.learner_config.replay.batch_size
        self.action_dim = self.env_config.action_spec.dim[0]
        self.obs_spec = self.env_config.obs_spec
        self.init_log_sig = algo.consts.init_log_sig
        # PPO parameters (ppo.py:88-118)
        self.ppo_mode = algo.ppo_mode
        self.if_rnn_policy = algo.rnn.if_rnn_policy
        self.horizon = algo.rnn.horizon
        self.lr_actor = algo.network.lr_actor
        self.lr_critic = algo.network.lr_critic
        self.epoch_policy = algo.consts.epoch_policy
        self.epoch_baseline = algo.consts.epoch_baseline
        self.kl_target = algo.consts.kl_target
        self.adjust_threshold = algo.consts.adjust_threshold
        self.reward_scale = algo.advantage.reward_scale
        self.kl_cutoff_coeff = algo.adapt_consts.kl_cutoff_coeff
        self.beta_init = algo.adapt_consts.beta_init
        self.beta_range = algo.adapt_consts.beta_range
        self.clip_range = algo.clip_consts.clip_range
        self.clip_epsilon_init = algo.clip_consts.clip_epsilon_init
        if self.ppo_mode == 'adapt':
            self.beta = self.beta_init
            self.eta = self.kl_cutoff_coeff
            self.beta_upper = self.beta_range[1]
            self.beta_lower = self.beta_range[0]
            self.beta_adjust_threshold = self.adjust_threshold
        elif self.ppo_mode == 'clip':
            self.clip_epsilon = self.clip_epsilon_init
            self.clip_adjust_threshold = self.adjust_threshold
            self.clip_upper = self.clip_range[1]
            self.clip_lower = self.clip_range[0]
        else:
            raise ValueError('ppo_mode must be "adapt" or "clip", got %r' % (self.ppo_mode,))
        # learning-rate annealing (ppo.py:121-125)
        anneal = algo.network.anneal
        self.min_lr = anneal.min_lr
        self.lr_update_frequency = anneal.lr_update_frequency
        self.frames_to_anneal = anneal.frames_to_anneal
        num_updates = int(self.frames_to_anneal / self.learner_config.parameter_publish.exp_interval)
        if anneal.lr_scheduler != 'LinearWithMinLR':
            raise ValueError('unknown lr_scheduler %r' % (anneal.lr_scheduler,))

        self.exp_counter = 0
        self.kl_record = []

        mk = dict(obs_spec=self.obs_spec, action_dim=self.action_dim,
                  model_config=self.learner_config.model, use_cuda=True,
                  init_log_sig=self.init_log_sig, use_z_filter=self.use_z_filter,
                  if_pixel_input=self.env_config.get('pixel_input', False),
                  rnn_config=algo.rnn, device=self.device, kernels=self.K)
        self.model = PPOModel(**mk)
        self.ref_target_model = PPOModel(**mk)
        self.ref_target_model.update_target_params(self.model)          # ppo.py:151

        net = algo.network
        self.clip_actor_gradient = net.clip_actor_gradient
        self.actor_gradient_clip_value = net.actor_gradient_norm_clip
        self.clip_critic_gradient = net.clip_critic_gradient
        self.critic_gradient_clip_value = net.critic_gradient_norm_clip
        self.actor_regularization = net.actor_regularization
        self.critic_regularization = net.critic_regularization
        # torch.optim.Adam state (ppo.py:159-168), one flat buffer per group
        self.actor_exp_avg = torch.zeros_like(self.model.actor_flat)
        self.actor_exp_avg_sq = torch.zeros_like(self.model.actor_flat)
        self.critic_exp_avg = torch.zeros_like(self.model.critic_flat)
        self.critic_exp_avg_sq = torch.zeros_like(self.model.critic_flat)
        self.actor_lr_scheduler = LinearWithMinLR(self.lr_actor, num_updates,
                                                  self.lr_update_frequency, self.min_lr)
        self.critic_lr_scheduler = LinearWithMinLR(self.lr_critic, num_updates,
                                                   self.lr_update_frequency, self.min_lr)
        self.aggregator = MultistepAggregatorWithInfo(self.env_config.obs_spec,
                                                      self.env_config.action_spec)
        self.pd = DiagGauss(self.action_dim)
        self.cells = None
        # reward scale / RewardFilter (ppo.py:452-455) run inside the step: one launch (smx_reward_filter_f32)
        self.filter_rewards = bool(self.use_r_filter) or self.reward_scale != 1.0

        lcfg = self.session_config.learner
        # one rank: the whole step is ONE hipGraph; several ranks: graph segments between the
        # collectives (_SegmentedGraph), unless session_config.learner.graph_segments is off
        self.use_graph = bool(lcfg.get('use_hip_graph', True)) and self.device != 'cpu' \
            and (self.world_size == 1 or bool(lcfg.get('graph_segments', True)))
        # actor and critic epochs share launches (lock-step: the reference's two loops touch disjoint parameters).
        # 'fused_epochs' (default where the shapes allow): an epoch is three launches of the row-block kernels
        # (csrc/smx_epoch.hip) instead of nine layer launches.  (Rounds 2-4 also carried a two-stream schedule, the
        # two chains on two streams inside the graph, and the weight gradients + Adam as one launch: each measured
        # equal or slower, DESIGN.md 3.2, and was removed in round 5.)
        self.fused_epochs = bool(lcfg.get('fused_epochs', True))
        self._ws = None
        self._graphs = {}
        self._ctrl_host = None
        self.trace = None

    # ======================================================================================
    # device-resident control block (smx_ppo_ctrl_t)
    # ======================================================================================
    def _ensure_ctrl(self, ws):
        vals = [self.actor_lr_scheduler.get_lr()[0], self.critic_lr_scheduler.get_lr()[0],
                getattr(self, 'beta', 0.0), getattr(self, 'eta', 0.0),
                getattr(self, 'clip_epsilon', 0.0), self.kl_target,
                self.actor_gradient_clip_value if self.clip_actor_gradient else 0.0,
                self.critic_gradient_clip_value if self.clip_critic_gradient else 0.0,
                self.actor_regularization, self.critic_regularization]
        if self._ctrl_host != vals:
            ws.ctrl_f[:10].copy_(torch.tensor(vals, dtype=torch.float32))
            self._ctrl_host = vals

    def _sync_words(self):
        return (max(self.epoch_policy, self.epoch_baseline) + 1 + 3) & ~3

    # ======================================================================================
    # workspace
    # ======================================================================================
    def _workspace(self, B, N, D, A, pix_dtype=None):
        key = (B, N, D, A, pix_dtype)
        if self._ws is not None and self._ws.key == key:
            return self._ws
        dev, K = self.device, self.K
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        ws = types.SimpleNamespace()
        ws.key = key
        act, cri = self.model.actor, self.model.critic
        rnn = self.if_rnn_policy
        pixel = self.model.if_pixel
        stem = rnn or pixel
        E = N - self.horizon + 1 if rnn else 1           # ppo.py:398-400, 521-537
        ws.E = E
        Ep, Ev = self.epoch_policy, self.epoch_baseline
        # scalars block: ctrl | policy stats | in-launch counters of the fused forward + backward epochs | value stats | moments
        # (+ per launch and row block of the fused forward + backward epochs: an 8-byte slot)
        n_slots = 0 if (self.if_rnn_policy or self.model.if_pixel or self.world_size > 1) else \
            2 * (max(Ep, Ev) + 1) * ((B + 15) // 16)
        if n_slots > (1 << 19):
            # (the slots sit in the range epoch_prepare zeroes per learn, which the kernel bounds at 2^20 words: batches of
            # more than ~380 k sub-trajectories run the two-launch epochs, which need none)
            n_slots = 0
        n_sync = self._sync_words() + n_slots
        n_scal = L.CTRL_WORDS + (Ep + 1) * L.PS_STRIDE + n_sync + Ev * L.VS_STRIDE + 12 + 4
        ws.scal = torch.zeros(n_scal, device=dev, dtype=torch.float32)
        o = 0
        ws.ctrl_f = ws.scal[o:o + L.CTRL_WORDS]; o += L.CTRL_WORDS
        ws.ctrl_i = ws.ctrl_f.view(torch.int32)
        ws.pstats = ws.scal[o:o + (Ep + 1) * L.PS_STRIDE].view(Ep + 1, L.PS_STRIDE); o += (Ep + 1) * L.PS_STRIDE
        ws.sync = ws.scal[o:o + n_sync].view(torch.int32); o += n_sync       # one word per epoch launch, zeroed per learn
        ws.kl_slots = ws.sync[self._sync_words():self._sync_words() + n_slots].view(-1, 2 * ((B + 15) // 16)) \
            if n_slots else None
        ws.n_sync = n_sync
        ws.vstats = ws.scal[o:o + Ev * L.VS_STRIDE].view(Ev, L.VS_STRIDE); o += Ev * L.VS_STRIDE
        ws.adv_mom = ws.scal[o:o + 3]; o += 3
        ws.ret_mom = ws.scal[o:o + 3]; o += 3
        ws.fin = ws.scal[o + 2:o + 6]          # mean log_var, z-filter means (final_stats)
        # RewardFilter's {count, running_sum, running_sumsq} (reward_filter.py:28-31): in the statistics block,
        # so the reported reward mean needs no read-back of its own; survives a change of workspace
        ws.rf_state = ws.scal[o + 6:o + 9]
        if self._ws is not None:
            ws.rf_state.copy_(self._ws.rf_state)
        else:
            ws.rf_state.copy_(torch.tensor([1e-5, 0.0, 0.0]))
        if self.filter_rewards:
            ws.rew = torch.empty(B, N, device=dev, dtype=torch.float32)
            ws.rf_part = torch.zeros(K.reward_filter_partials(), device=dev, dtype=torch.float64)
            ws.rf_ticket = torch.zeros(1, device=dev, dtype=torch.int32)
            ws.rf_sums = torch.zeros(3, device=dev, dtype=torch.float32)
        ws.stop = ws.ctrl_i[L.C_STOP:L.C_STOP + 1]
        # stop flag + epochs_done + reserved words + the policy statistics rows: one contiguous run
        ws.zero_block = ws.scal[L.C_STOP:L.CTRL_WORDS + (Ep + 1) * L.PS_STRIDE + n_sync]
        # keep the optimiser step counters of a previous workspace
        if self._ws is not None:
            ws.ctrl_i[L.C_STEP_ACTOR:L.C_STEP_CRITIC + 1].copy_(
                self._ws.ctrl_i[L.C_STEP_ACTOR:L.C_STEP_CRITIC + 1])
        self._ctrl_host = None
        # critic pass + GAE
        ws.packed = None if stem else f(K.mlp3_packed_numel(cri))
        ws.vals = f(B * (N + 1))
        # fused-kernel tail split (see _enqueue_gae): rounds of 128-row workgroups over the CUs
        n_cu = 256
        if dev != 'cpu':
            n_cu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        rounds = lambda rows: -(-(-(-rows // 128)) // n_cu)  # noqa: E731
        ws.split_tail = bool(self.session_config.learner.get('split_critic_tail', True)) and \
            rounds(B * (N + 1)) > rounds(B * N) and not stem
        ws.xnext = f(B, D)
        ws.h1t, ws.h2t = f(B, cri.H1), f(B, cri.H2)      # critic over the obs_next rows
        ws.adv = f(B * E)
        ws.ret = f(B * E)
        idx = torch.tensor(range(N), dtype=torch.float32)
        ws.gpow = torch.pow(self.gamma, idx).to(dev)     # ppo.py:372-374, built as the reference does
        ws.lpow = torch.pow(self.lam, idx).to(dev)
        ws.mom_parts = f(self.world_size, 3)
        # epochs
        rows = B * E
        ws.rows = rows
        Dx = self.model.stem_in       # D, or D + cnn_feature_dim with pixel observations
        ws.xn = f(rows, Dx)           # z-filtered step-0 observations [| CNN features] (model filter)
        ws.xr = f(rows, Dx)           # same through the reference-policy filter / stem
        ws.h1a, ws.h2a, ws.mean = f(rows, act.H1), f(rows, act.H2), f(rows, A)
        ws.h1r, ws.h2r, ws.ref_mean = f(rows, act.H1), f(rows, act.H2), f(rows, A)
        ws.ref_pol = f(rows, 2 * A)
        ws.h1c, ws.h2c, ws.vpred = f(rows, cri.H1), f(rows, cri.H2), f(rows)
        ws.g_surr, ws.g_kl, ws.dz3a = f(rows, A), f(rows, A), f(rows, A)
        ws.dz2a, ws.dz1a = f(rows, act.H2), f(rows, act.H1)
        ws.dz3c, ws.dz2c, ws.dz1c = f(rows), f(rows, cri.H2), f(rows, cri.H1)
        if stem:
            R1 = B * (N + 1)
            ws.h0 = ws.c0 = ws.cnn_it = ws.cnn_gae = ws.frames_it = None
            ws.low_it = f(rows, D)                    # obs[:, :E] (raw: also feeds z_update)
            ws.act_it, ws.beh_it = f(rows, A), f(rows, 2 * A)
            ws.ztmp = f(R1, D)
            ws.lcat, ws.xcat = f(B, N + 1, D), f(R1, Dx)     # critic pass over cat(obs, obs_next)
            ws.h1G, ws.h2G = f(R1, cri.H1), f(R1, cri.H2)
            n_sk = max(K.mlp3_backward_ws_floats(n, rows) for n in (act, cri))
            ws.mlp_sk = f(n_sk) if n_sk else None          # split-K partials of the MLP weight gradients over B x T rows
            # scratch for the packed weights of the fused many-row forward (K.mlp3_forward(pack=...): one launch instead
            # of three layer GEMMs from FUSED_ROWS_MIN rows on; the shapes it does not take keep the layered path)
            try:
                ws.pack_stem = f(max(K.mlp3_packed_numel(n) for n in (act, cri)))
            except Exception:
                ws.pack_stem = None
            n_z = K.zfilter_update_ws_floats(rows, D)
            ws.zscratch = f(n_z) if n_z else None          # the z-update's chunk sums over B * E rows (K.zfilter_update)
            n_pt = max(K.mlp3_dgrad_rows_ws_floats(n) for n in (act, cri))
            ws.packT_stem = f(n_pt) if n_pt else None    # ... and of the fused many-row data gradients (K.mlp3_backward)
            if rnn:
                # LSTM stem (ppo_net.py:143-152): sequence buffers for the epoch passes (T = E)
                # and for the critic pass (T = N + 1, ppo.py:376-386)
                F = self.model.rnn_hidden
                nl = self.model.rnn_layers
                ws.h0L, ws.c0L = f(nl, B, F), f(nl, B, F)          # agent-side state of every layer
                ws.h0, ws.c0 = ws.h0L[0], ws.c0L[0]
                ws.gates, ws.lo, ws.cs, ws.hp, ws.dlo = f(rows, 4 * F), f(rows, F), f(rows, F), f(rows, F), f(rows, F)
                ws.gatesG, ws.loG, ws.csG = f(R1, 4 * F), f(R1, F), f(R1, F)
                # stacked layers above the first: their own sequence buffers (layer l reads layer l-1's lo)
                ws.upper = [types.SimpleNamespace(gates=f(rows, 4 * F), lo=f(rows, F), cs=f(rows, F),
                                                  hp=f(rows, F), dlo=f(rows, F)) for _ in range(nl - 1)]
                ws.loG2 = f(R1, F) if nl > 1 else None
                n_sk = max(K.lstm_backward_ws_floats(r, B, E) for r in self.model.rnns)
                ws.lstm_sk = f(n_sk) if n_sk else None
            if pixel:
                # CNN stem (builders.py:8-33): frames stay in their source dtype (uint8 from the
                # cameras); the critic pass runs the stem in chunks to bound the patch matrices
                from surreal_amd.model.cnn_stem import CnnStem
                cnn = self.model.cnn
                cam = (cnn.C, cnn.H, cnn.W)
                ws.frames_it = torch.empty((rows,) + cam, device=dev, dtype=pix_dtype)
                ws.fcat = torch.empty((B, N + 1) + cam, device=dev, dtype=pix_dtype)
                ws.dxn = torch.zeros(rows, Dx, device=dev)
                chunk = int(self.session_config.learner.get('cnn_chunk_frames', 2048))
                ws.cnn_it = CnnStem.workspace(cnn, rows, dev, backward=True)
                ws.cnn_it.sk = self.model._cnn_stem.splitk_workspace(cnn, rows, dev)
                ws.cnn_gae = ws.cnn_it if rows >= min(chunk, R1) else \
                    CnnStem.workspace(cnn, min(chunk, R1), dev, backward=False)
        else:
            # transposed copies [features, rows] feeding the weight-gradient GEMMs (K-contiguous)
            # (row stride padded off the power of two: all 32 rows of a fragment load would
            # otherwise land on one cache set / memory channel)
            ldT = rows + 16
            ft = lambda n: torch.zeros(n, ldT, device=dev, dtype=torch.float32)[:, :rows]  # noqa: E731
            ws.xnT = ft(D)
            ws.h1aT, ws.h2aT, ws.dz3aT = ft(act.H1), ft(act.H2), ft(A)
            ws.dz2aT, ws.dz1aT = ft(act.H2), ft(act.H1)
            ws.h1cT, ws.h2cT, ws.dz3cT = ft(cri.H1), ft(cri.H2), ft(1)
            ws.dz2cT, ws.dz1cT = ft(cri.H2), ft(cri.H1)
        # one buffer for both groups' gradients: a data-parallel lock-step epoch all-reduces it once
        n_a, n_c = self.model.actor_flat.numel(), self.model.critic_flat.numel()
        ws.nblk_p = K.loss_blocks(rows)
        # fused row-block epochs: plain MLP policy, shapes the kernels take; on several ranks the
        # paired-epoch schedule with one collective per epoch (epoch_policy == epoch_baseline)
        ws.fused = (self.fused_epochs and not stem and
                    (self.world_size == 1 or self.epoch_policy == self.epoch_baseline) and
                    K.epoch_supported(act, cri))
        # ... and an updating epoch's forward + loss + data gradients as ONE launch (single rank; several ranks need the
        # all-reduce between them)
        # In adapt mode every actor workgroup of that launch WAITS inside it for the KL sums of all the others, so all
        # of them must be resident at once.  The C side checks the launch against the CU count; what it cannot see is
        # who else is on the device: kernels of another stream (actors one rollout ahead of the learner) or of
        # another process hold CUs for as long as they run, and the wait is bounded (0.25 s, then the learn fails
        # loudly).  A learner that may share the device says so -- session_config.learner.exclusive_device = False
        # (set it on shared GPUs) -- and runs the two-launch form: same results, one more launch per epoch.
        ws.fb = (ws.fused and self.world_size == 1 and bool(self.session_config.learner.get('fused_fwdbwd', True)) and
                 bool(self.session_config.learner.get('exclusive_device', True)) and not getattr(self, '_fb_timed_out', False) and
                 n_slots > 0 and K.epoch_fwdbwd_supported(act, cri))
        vblocks = K.epoch_blocks if ws.fused else K.value_loss_blocks     # value-loss moments per 16 / 256 rows
        ws.nblk_v = vblocks(rows)
        # single rank: GAE + normalisation and the end-of-learn statistics are one launch each
        ws.merged_tail = ws.fused and self.world_size == 1
        ws.ticket = torch.zeros(2, dtype=torch.int32, device=dev)
        if ws.fused:
            assert ws.nblk_p == K.epoch_blocks(rows)
            # the weights in the forward kernel's fragment order (model, critic, reference policy)
            ws.pk_actor = torch.zeros(K.epoch_packed_numel(act), device=dev)
            ws.pk_critic = torch.zeros(K.epoch_packed_numel(cri), device=dev)
            ws.pk_ref = torch.zeros(K.epoch_packed_numel(act), device=dev)
        ws.pstride = 8 + 2 * A
        # batch means are over the GLOBAL batch: ranks may hold different numbers of sub-trajectories
        # (B not divisible by the world size), so the totals are exchanged once per workspace.  A
        # rank's batch shape may therefore only change in a learn() where every rank's does.
        ws.n_total, ws.B_total, nblk_p_all = rows, B, ws.nblk_p
        if self.world_size > 1:
            mine = torch.tensor([rows, B], dtype=torch.int64, device=dev)
            every = torch.empty(2 * self.world_size, dtype=torch.int64, device=dev)
            self._dist.all_gather_into_tensor(every, mine)
            every = every.view(-1, 2).tolist()
            ws.n_total, ws.B_total = sum(r for r, _ in every), sum(b for _, b in every)
            ws.nblk_v = max(vblocks(r) for r, _ in every)
            nblk_p_all = max(K.loss_blocks(r) for r, _ in every)
        ws.dp_epoch = self.world_size > 1 and not stem
        ws.tail_deferred = ws.dp_epoch and self.epoch_policy >= self.epoch_baseline
        if ws.dp_epoch:
            # [surrogate share of the actor gradient | critic gradient | KL share (adapt) | loss
            # partial rows]: everything an epoch exchanges, in ONE all-reduce (_enqueue_lockstep_epochs)
            al = lambda n: (n + 3) & ~3  # noqa: E731
            adapt = self.ppo_mode != 'clip'
            off_k = al(n_a + n_c)
            off_p = off_k + (al(act.numel) if adapt else 0)
            ws.ar = torch.zeros(off_p + nblk_p_all * ws.pstride, device=dev)
            ws.grads_all = ws.ar[:n_a + n_c]
            ws.grads_k = ws.ar[off_k:off_k + act.numel] if adapt else None
            # (rows past this rank's own blocks must be 0 when the all-reduce reads them: see _clear_foreign_partials)
            ws.ppart_ar = ws.ar[off_p:].view(nblk_p_all, ws.pstride)
            if adapt:
                ws.dz3k, ws.dz2k, ws.dz1k = f(rows, A), f(rows, act.H2), f(rows, act.H1)
                ws.dz3kT, ws.dz2kT, ws.dz1kT = ft(A), ft(act.H2), ft(act.H1)
                ws.sumsq_k = torch.zeros(K.mlp3_backward_partials(act), device=dev)
            # what the end of a learn exchanges, packed into one all-gather (_enqueue_tail_exchange):
            # [loss sums of the final policy pass | value-loss moments of all epochs | return
            #  moments | z-filter column sums]
            nz = 2 * D + 1 if self.use_z_filter else 0
            cuts = np.cumsum([0, ws.pstride, Ev * ws.nblk_v * 8, 3, nz]).tolist()
            ws.tail_cuts = cuts
            ws.tail_pack = torch.zeros(cuts[-1], device=dev)
            ws.tail_gather = f(self.world_size, cuts[-1])
            ws.tp_ppart = ws.tail_pack[:cuts[1]].view(1, ws.pstride)
            ws.vpart_loc_all = ws.tail_pack[cuts[1]:cuts[2]].view(Ev, ws.nblk_v, 8)
            ws.tp_ret = ws.tail_pack[cuts[2]:cuts[3]]
            ws.zsum = f(nz) if nz else None
        else:
            ws.grads_all = torch.zeros(n_a + n_c, device=dev)
        ws.grads_a, ws.grads_c = ws.grads_all[:n_a], ws.grads_all[n_a:]
        ws.ppart = f(ws.nblk_p, ws.pstride)
        ws.ppart_sum = f(1, ws.pstride)
        ws.ppart_fold = f(64, ws.pstride)
        # Stem policies on several ranks, clip mode: ONE exchange per policy epoch.  Nothing in the clip gradient depends on
        # the batch (dz3 = g_surr / n), so the loss sums travel WITH the gradient -- [actor group's gradient | loss sums] is
        # one buffer, one all-reduce -- and the statistics, log_var's gradient and the KL early-exit flag are formed from
        # the global sums behind it, in front of the optimiser launch that honours the flag (_stem_policy_update).  Adapt
        # mode keeps two: its KL coefficient needs the global KL BEFORE the backward pass, and the two-right-hand-side
        # form that avoids that would run the stem's backward twice.
        ws.stem_one_exchange = bool(stem and self.world_size > 1 and self.ppo_mode == 'clip' and
                                    self.session_config.learner.get('stem_one_exchange', True))
        if ws.stem_one_exchange:
            ws.ar_pol = torch.zeros(((n_a + 3) & ~3) + ws.pstride, device=dev)
            ws.grads_a = ws.ar_pol[:n_a]
            ws.ppart_sum = ws.ar_pol[(n_a + 3) & ~3:].view(1, ws.pstride)
        # partial rows a rank does not fill stay zero (count 0: skipped by the merge)
        ws.vpart = torch.zeros(Ev, self.world_size * ws.nblk_v, 8, device=dev)
        ws.vpart_local = torch.zeros(ws.nblk_v, 8, device=dev)
        if not ws.dp_epoch:
            ws.vpart_loc_all = torch.zeros(Ev, ws.nblk_v, 8, device=dev)   # lock-step: gathered once per learn
        ws.vgather = f(self.world_size, Ev, ws.nblk_v, 8)
        ws.np_a = K.mlp3_backward_partials(act)
        ws.np_c = K.mlp3_backward_partials(cri)
        ws.sumsq_a = torch.zeros(max(ws.np_a + 1, K.sumsq_blocks(ws.grads_a.numel())), device=dev)
        ws.sumsq_c = torch.zeros(max(ws.np_c, K.sumsq_blocks(ws.grads_c.numel())), device=dev)
        if self.use_z_filter:
            ws.zdelta = ws.tail_pack[ws.tail_cuts[3]:] if ws.dp_epoch else torch.zeros(2 * D + 1, device=dev)
        ws.xerr = ws.ctrl_i[L.C_XCHG_ERR:L.C_XCHG_ERR + 1]
        if self.world_size > 1:
            self._setup_peer_exchange(ws, n_a + n_c)
        self._ws = ws
        self._graphs = {}
        return ws

    def _setup_peer_exchange(self, ws, n_grads):
        """several ranks on one node: the fp32 exchanges of a learn() as kernels over IPC-mapped peer buffers
        (surreal_amd.distributed.PeerExchange) -- set up and SELF-CHECKED once, collectively; any failure leaves
        the process group (RCCL) in place.  session_config.learner.peer_exchange = False keeps RCCL."""
        d = self._dist
        d.err_word = ws.xerr
        want = bool(self.session_config.learner.get('peer_exchange', True)) and self.device != 'cpu'
        need = max(ws.ar.numel() if ws.dp_epoch else n_grads, 64)
        if getattr(ws, 'tail_gather', None) is not None:
            need = max(need, ws.tail_gather.numel())
        if not want:
            return
        if d.exchange is not None and d.exchange.capacity >= need:
            return
        if d.exchange is not None:
            d._d.barrier()
            d.exchange.close()
            d.exchange = None
        from surreal_amd.distributed.peer_exchange import PeerExchange
        d.exchange = PeerExchange.create(d._d, need, timeout_s=float(self.session_config.learner.get(
            'peer_exchange_timeout_s', 5.0)))
        self.exchange_kind = 'peer buffers (%s)' % d.exchange.check_message if d.exchange is not None else 'process group'

    # ======================================================================================
    # batch handling
    # ======================================================================================
    def _to_dev(self, x):
        if torch.is_tensor(x):
            return x.to(self.device, torch.float32)
        return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device)

    def _to_dev_pixels(self, x):
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        if x.dtype != torch.uint8:
            x = x.to(torch.float32)
        return x.to(self.device)

    def _preprocess_batch_ppo(self, batch):
        """numpy -> device fp32 tensors (ppo.py:420-484); device tensors pass through"""
        obs, obs_next = batch['obs'], batch['obs_next']
        for modality in obs:
            for key in obs[modality]:
                # camera frames stay uint8 (the CNN stem's patch kernel applies x / 255 while it
                # reads them; the reference converts the whole batch to fp32 first, ppo.py:436-441)
                conv = self._to_dev_pixels if modality == 'pixel' else self._to_dev
                obs[modality][key] = conv(obs[modality][key])
                obs_next[modality][key] = conv(obs_next[modality][key])
        batch['actions'] = self._to_dev(batch['actions'])
        # (reward_scale and the RewardFilter of ppo.py:452-455 are applied by the step itself:
        # _enqueue_reward_filter, one launch inside the captured graph)
        batch['rewards'] = self._to_dev(batch['rewards'])
        batch['dones'] = self._to_dev(batch['dones'])
        if batch.get('persistent_infos') is not None:
            batch['persistent_infos'] = [self._to_dev(x) for x in batch['persistent_infos']]
        if batch.get('onetime_infos') is not None:
            batch['onetime_infos'] = [self._to_dev(x) for x in batch['onetime_infos']]
        return batch

    def _flat_obs(self, obs):
        parts = [obs['low_dim'][k] for k in obs['low_dim'].keys()]
        x = parts[0] if len(parts) == 1 else torch.cat(parts, -1)
        return x.contiguous()

    # ======================================================================================
    # pieces of _optimize (each is a short chain of kernel launches, no host sync)
    # ======================================================================================
    def _enqueue_gae(self, ws, obs, obs_next, rewards, dones):
        """critic over all steps + windowed GAE + normalisation (ppo.py:355-418)"""
        tail = self._enqueue_critic_pass(ws, obs, obs_next)
        if tail is not None:
            self.K.mlp3_forward_multi([tail])
        self._enqueue_gae_from_values(ws, obs, rewards, dones)

    def _enqueue_critic_pass(self, ws, obs, obs_next, filter_tail=True):
        """the critic over all steps.  Returns the forward job of the obs_next rows when they are
        left to the layered kernels (the caller launches it, alone or together with other jobs;
        filter_tail=False: the caller also z-filters those rows into ws.xnext)."""
        K, m = self.K, self.model
        B, N, D = obs.shape
        zm = zs = None
        if self.use_z_filter:            # repack + z-filter statistics: one launch
            K.mlp3_pack_zstats(m.critic, ws.packed, m.z_filter)
            zm, zs = m.z_filter._mean, m.z_filter._std
        else:
            K.mlp3_pack(m.critic, ws.packed)
        if not ws.split_tail:
            K.mlp3_forward_fused(ws.packed, m.critic, obs, obs_next, zm, zs, ws.vals, L.SMX_ACT_NONE)
            return None
        # B*(N+1) rows would need one more (nearly empty) round of workgroups over the chip
        # than B*N: the fused kernel takes the B*N step rows, the B obs_next rows go through
        # the layered small-batch kernels (same arithmetic, tests check both against the oracle)
        K.mlp3_forward_fused(ws.packed, m.critic, obs, None, zm, zs, ws.vals[:B * N], L.SMX_ACT_NONE)
        if not filter_tail:
            pass
        elif self.use_z_filter:
            K.zfilter_forward(obs_next[:, 0, :], zm, zs, ws.xnext)
        else:
            ws.xnext.copy_(obs_next[:, 0, :])
        return dict(net=m.critic, x=ws.xnext, h1=ws.h1t, h2=ws.h2t, out=ws.vals[B * N:].view(B, 1),
                    act=L.SMX_ACT_NONE)

    def _enqueue_gae_from_values(self, ws, obs, rewards, dones):
        K = self.K
        B, N, D = obs.shape
        if ws.merged_tail and self.norm_adv:
            K.gae_norm(ws.vals[:B * N] if ws.split_tail else ws.vals, rewards, dones, ws.gpow, ws.lpow, self.gamma,
                       self.gamma ** N, B, N, N, ws.adv, ws.ret, ws.adv_mom, 1e-4, ws.ticket[0:1],
                       values_tail=ws.vals[B * N:] if ws.split_tail else None)
            return
        if ws.split_tail:
            K.gae(ws.vals[:B * N], rewards, dones, ws.gpow, ws.lpow, self.gamma, self.gamma ** N,
                  B, N, N, ws.adv, ws.ret, values_tail=ws.vals[B * N:])
        else:
            K.gae(ws.vals, rewards, dones, ws.gpow, ws.lpow, self.gamma, self.gamma ** N, B, N, N,
                  ws.adv, ws.ret)
        if self.norm_adv:
            K.moments(ws.adv, ws.adv_mom)
            if self.world_size > 1:
                self._dist.all_gather_into_tensor(ws.mom_parts.view(-1), ws.adv_mom.clone())
                K.moments_merge(ws.mom_parts, ws.adv_mom)
            K.adv_normalize(ws.adv, ws.adv_mom, 1e-4)

    def _enqueue_lockstep_epochs(self, ws, actions0, behave0, first_extra=(), after_first=None):
        """Policy epoch e and value epoch e advance together: the reference runs the two loops
        one after the other (ppo.py:541-562) but they touch disjoint parameters, so the layer-l
        GEMMs of the actor and of the critic share one launch (smx_mlp3_*_multi_f32).  The KL
        early exit only masks the actor's share (its jobs carry the device stop flag)."""
        K, m = self.K, self.model
        Ep, Ev = self.epoch_policy, self.epoch_baseline
        mode = L.SMX_PPO_CLIP if self.ppo_mode == 'clip' else L.SMX_PPO_ADAPT
        A = self.action_dim
        W = self.world_size
        n_total = ws.n_total
        aj = dict(net=m.actor, x=ws.xn, h1=ws.h1a, h2=ws.h2a, out=ws.mean, act=L.SMX_ACT_TANH,
                  dz3=ws.dz3a, dz2=ws.dz2a, dz1=ws.dz1a, grads=ws.grads_a, sumsq=ws.sumsq_a,
                  stop=ws.stop, xT=ws.xnT, h1T=ws.h1aT, h2T=ws.h2aT, dz3T=ws.dz3aT, dz2T=ws.dz2aT,
                  dz1T=ws.dz1aT)
        cj = dict(net=m.critic, x=ws.xn, h1=ws.h1c, h2=ws.h2c, out=ws.vpred.view(-1, 1),
                  act=L.SMX_ACT_NONE, dz3=ws.dz3c.view(-1, 1), dz2=ws.dz2c, dz1=ws.dz1c,
                  grads=ws.grads_c, sumsq=ws.sumsq_c, xT=ws.xnT, h1T=ws.h1cT, h2T=ws.h2cT,
                  dz3T=ws.dz3c, dz2T=ws.dz2cT, dz1T=ws.dz1cT)     # OUT = 1: dz3^T is dz3 itself
        for e in range(max(Ep + 1, Ev)):
            pol_f, pol_u, val = e <= Ep, e < Ep, e < Ev
            K.mlp3_forward_multi(([aj] if pol_f else []) + ([cj] if val else []) +
                                 (list(first_extra) if e == 0 else []))
            if e == 0 and after_first is not None:
                after_first()
            if ws.dp_epoch and pol_u and val:
                self._enqueue_dp_epoch(ws, e, mode, aj, cj, actions0, behave0)
                continue
            if pol_f and W == 1:
                # one ABI call: policy loss and value loss share a launch, then the policy finalize
                K.epoch_losses(mode, ws.mean, m.log_var.view(-1), actions0, behave0, ws.ref_pol, ws.adv,
                               ws.ctrl_f, ws.g_surr, ws.g_kl, ws.ppart, e > 0, pol_u, ws.dz3a,
                               ws.grads_a[m.actor.numel:m.actor.numel + A],
                               ws.sumsq_a[ws.np_a:ws.np_a + 1], ws.pstats[e], dz3_t=ws.dz3aT,
                               values=ws.vpred if val else None, returns=ws.ret, v_dz3=ws.dz3c,
                               v_partials=ws.vpart[e] if val else None)
            elif pol_f and ws.tail_deferred and not pol_u and not val:
                # the final, forward-only policy pass: its loss sums travel with the end-of-learn
                # exchange (_enqueue_tail_exchange), which also runs its finalize
                K.policy_loss(mode, ws.mean, m.log_var.view(-1), actions0, behave0, ws.ref_pol,
                              ws.adv, ws.ctrl_f, ws.g_surr, ws.g_kl, ws.ppart)
                torch.sum(ws.ppart, 0, keepdim=True, out=ws.tp_ppart)
                continue
            elif pol_f:
                K.policy_loss(mode, ws.mean, m.log_var.view(-1), actions0, behave0, ws.ref_pol,
                              ws.adv, ws.ctrl_f, ws.g_surr, ws.g_kl, ws.ppart)
                part, nblk = ws.ppart, ws.nblk_p
                if W > 1:
                    torch.sum(ws.ppart, 0, keepdim=True, out=ws.ppart_sum)
                    self._dist.all_reduce(ws.ppart_sum)
                    part, nblk = ws.ppart_sum, 1
                K.policy_finalize(mode, part, nblk, ws.g_surr, ws.g_kl, m.log_var.view(-1), n_total,
                                  ws.ctrl_f, e > 0, pol_u, ws.dz3a,
                                  ws.grads_a[m.actor.numel:m.actor.numel + A],
                                  ws.sumsq_a[ws.np_a:ws.np_a + 1], ws.pstats[e], dz3_t=ws.dz3aT)
            if val and not (pol_f and W == 1):
                # the value-loss partial sums only feed statistics: every epoch's stay local and
                # are gathered once at the end of the learn
                K.value_loss(ws.vpred, ws.ret, n_total, ws.dz3c,
                             ws.vpart_loc_all[e] if W > 1 else ws.vpart[e], ws.ctrl_f, True)
            if pol_u or val:
                K.mlp3_backward_multi(([aj] if pol_u else []) + ([cj] if val else []))
            np_a, np_c = ws.np_a + 1, ws.np_c
            if W > 1:
                if pol_u and val:
                    # ONE all-reduce for both groups.  log_var's gradient was built from the
                    # all-reduced loss partials and is global already: exactly one copy may enter
                    # the sum (exact for any world size)
                    if self.rank != 0:
                        ws.grads_a[m.actor.numel:m.actor.numel + A].zero_()
                    self._dist.all_reduce(ws.grads_all)
                elif pol_u:
                    self._dist.all_reduce(ws.grads_a[:m.actor.numel])
                elif val:
                    self._dist.all_reduce(ws.grads_c)
                if pol_u:
                    K.sumsq_partials(ws.grads_a, ws.sumsq_a)
                    np_a = K.sumsq_blocks(ws.grads_a.numel())
                if val:
                    K.sumsq_partials(ws.grads_c, ws.sumsq_c)
                    np_c = K.sumsq_blocks(ws.grads_c.numel())
            if pol_u and val:        # both groups step in one launch
                K.clip_adam_pair((m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                                  ws.sumsq_a, np_a, True, ws.pstats[e, L.PS_GRADNORM:L.PS_GRADNORM + 1]),
                                 (m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                                  ws.sumsq_c, np_c, False, ws.vstats[e, L.VS_GRADNORM:L.VS_GRADNORM + 1]),
                                 ws.ctrl_f)
            elif pol_u:
                K.clip_adam(m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                            ws.sumsq_a, np_a, ws.ctrl_f, 0, True,
                            ws.pstats[e, L.PS_GRADNORM:L.PS_GRADNORM + 1])
            elif val:
                K.clip_adam(m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                            ws.sumsq_c, np_c, ws.ctrl_f, 1, False,
                            ws.vstats[e, L.VS_GRADNORM:L.VS_GRADNORM + 1])
        if ws.tail_deferred:
            return
        if W > 1:
            self._dist.all_gather_into_tensor(ws.vgather.view(-1), ws.vpart_loc_all.view(-1))
            ws.vpart.view(Ev, W, ws.nblk_v, 8).copy_(ws.vgather.permute(1, 0, 2, 3))
        K.value_finalize(ws.vpart, Ev, ws.vpart.shape[1], ws.vstats, L.VS_STRIDE)

    def _clear_foreign_partials(self, ws):
        """The all-reduce sums ws.ar in place, so afterwards EVERY loss-partial row holds the global sum --
        also the rows past this rank's own blocks, which its loss launch never rewrites (ranks may hold
        different numbers of sixteen-row blocks).  Those rows go back to zero before the next epoch's
        exchange, or the previous epoch's sums would be added again."""
        if ws.nblk_p < ws.ppart_ar.shape[0]:
            ws.ppart_ar[ws.nblk_p:].zero_()

    def _enqueue_fused_epochs(self, ws, actions0, behave0, ref_job, tail, gae):
        """The lock-step epochs on the row-block kernels (csrc/smx_epoch.hip): per epoch
        [forward + losses] -> [batch means, KL coefficient / early exit, data gradients] ->
        [all weight gradients] -> [clip-norm + Adam of both groups]: four dependent launches where
        the layered schedule takes nine.  Same arithmetic contract as _enqueue_lockstep_epochs."""
        K, m = self.K, self.model
        Ep, Ev = self.epoch_policy, self.epoch_baseline
        mode = L.SMX_PPO_CLIP if self.ppo_mode == 'clip' else L.SMX_PPO_ADAPT
        A = self.action_dim
        n_total = ws.n_total
        # the reference policy and the critic's obs_next rows: one forward-only launch, then GAE
        pre = [dict(net=ref_job['net'], packed=ws.pk_ref, x=ref_job['x'], out=ref_job['out'], act=L.SMX_ACT_TANH)]
        if tail is not None:
            pre.append(dict(net=tail['net'], packed=ws.pk_critic, x=tail['x'], out=tail['out'], act=L.SMX_ACT_NONE))
        K.epoch_forward(pre, None, ws.ctrl_f, n_total)
        gae()
        aj = dict(net=m.actor, packed=ws.pk_actor, x=ws.xn, act=L.SMX_ACT_TANH, loss='policy', stop=ws.stop, h1T=ws.h1aT,
                  h2T=ws.h2aT, dz3T=ws.dz3aT, dz2T=ws.dz2aT, dz1T=ws.dz1aT, xT=ws.xnT, grads=ws.grads_a,
                  sumsq=ws.sumsq_a)
        cj = dict(net=m.critic, packed=ws.pk_critic, x=ws.xn, act=L.SMX_ACT_NONE, loss='value', h1T=ws.h1cT, h2T=ws.h2cT,
                  dz3=ws.dz3c, dz3T=ws.dz3c, dz2T=ws.dz2cT, dz1T=ws.dz1cT, xT=ws.xnT, grads=ws.grads_c,
                  sumsq=ws.sumsq_c)         # OUT = 1: dz3^T is dz3 itself
        dp = ws.dp_epoch                     # several ranks: one all-reduce per epoch (see _enqueue_dp_epoch)
        adapt = mode == L.SMX_PPO_ADAPT

        def loss_args(e):
            return dict(mode=mode, rows=ws.rows, log_var=m.log_var.view(-1), actions=actions0, behave=behave0,
                        ref=ws.ref_pol, adv=ws.adv, g_surr=ws.g_surr, g_kl=ws.g_kl,
                        partials=ws.ppart_ar if dp and e < Ep else ws.ppart,
                        check_stop=e > 0, will_update=e < Ep,
                        dlogvar=ws.grads_a[m.actor.numel:m.actor.numel + A],
                        dlogvar_sumsq=ws.sumsq_a[ws.np_a:ws.np_a + 1], stats=ws.pstats[min(e, Ep)],
                        returns=ws.ret, v_dz3=ws.dz3c,
                        v_partials=(ws.vpart_loc_all if dp else ws.vpart)[min(e, Ev - 1)], v_will_update=True)

        if dp:
            assert Ep == Ev and ws.tail_deferred
            rhs = [dict(aj, loss='rhs_surr')]
            if adapt:
                rhs.append(dict(aj, loss='rhs_kl', dz3T=ws.dz3kT, dz2T=ws.dz2kT, dz1T=ws.dz1kT, grads=ws.grads_k,
                                sumsq=ws.sumsq_k))
            for e in range(Ep + 1):
                loss = loss_args(e)
                if e == Ep:
                    # the final, forward-only policy pass: its loss sums travel with the end-of-learn
                    # exchange (_enqueue_tail_exchange), which also runs its finalize
                    K.epoch_forward([aj], loss, ws.ctrl_f, n_total)
                    torch.sum(ws.ppart, 0, keepdim=True, out=ws.tp_ppart)
                    break
                self._clear_foreign_partials(ws)
                K.epoch_forward([aj, cj], loss, ws.ctrl_f, n_total)
                K.epoch_backward(rhs + [cj], loss, ws.ctrl_f, n_total)
                K.mlp3_wgrad_multi(rhs + [cj])
                self._dist.all_reduce(ws.ar)
                K.epoch_combine(mode, ws.ppart_ar, ws.ppart_ar.shape[0], n_total, m.log_var.view(-1), ws.ctrl_f,
                                e > 0, True, ws.pstats[e], ws.grads_a, ws.grads_k if adapt else None,
                                m.actor.numel, ws.sumsq_a, ws.grads_c, ws.sumsq_c)
                K.clip_adam_pair((m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                                  ws.sumsq_a, K.sumsq_blocks(ws.grads_a.numel()), True,
                                  ws.pstats[e, L.PS_GRADNORM:L.PS_GRADNORM + 1]),
                                 (m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                                  ws.sumsq_c, K.sumsq_blocks(ws.grads_c.numel()), False,
                                  ws.vstats[e, L.VS_GRADNORM:L.VS_GRADNORM + 1]),
                                 ws.ctrl_f, pack=((m.actor, ws.pk_actor), (m.critic, ws.pk_critic)))
            return

        for e in range(max(Ep + 1, Ev)):
            pol_f, pol_u, val = e <= Ep, e < Ep, e < Ev
            loss = loss_args(e)
            if ws.fb and (pol_u or not pol_f) and (pol_u or val):
                # forward + loss + data gradients of every job of the epoch in ONE launch (smx_epoch_fwdbwd_f32)
                bj = ([aj] if pol_u else []) + ([cj] if val else [])
                K.epoch_fwdbwd(bj, loss, ws.ctrl_f, n_total, ws.sync[e:e + 1], ws.kl_slots[e])
                K.mlp3_wgrad_multi(bj)
            else:
                K.epoch_forward(([aj] if pol_f else []) + ([cj] if val else []), loss, ws.ctrl_f, n_total)
                if pol_f and not pol_u:
                    K.epoch_backward([aj], loss, ws.ctrl_f, n_total)       # statistics + early exit only
                if pol_u or val:
                    bj = ([aj] if pol_u else []) + ([cj] if val else [])
                    K.epoch_backward(bj, loss, ws.ctrl_f, n_total)
                    K.mlp3_wgrad_multi(bj)
            np_a, np_c = ws.np_a + 1, ws.np_c
            if pol_u and val:        # both groups step in one launch
                K.clip_adam_pair((m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                                  ws.sumsq_a, np_a, True, ws.pstats[e, L.PS_GRADNORM:L.PS_GRADNORM + 1]),
                                 (m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                                  ws.sumsq_c, np_c, False, ws.vstats[e, L.VS_GRADNORM:L.VS_GRADNORM + 1]),
                                 ws.ctrl_f, pack=((m.actor, ws.pk_actor), (m.critic, ws.pk_critic)))
            elif pol_u:
                K.clip_adam(m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                            ws.sumsq_a, np_a, ws.ctrl_f, 0, True,
                            ws.pstats[e, L.PS_GRADNORM:L.PS_GRADNORM + 1])
            elif val:
                K.clip_adam(m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                            ws.sumsq_c, np_c, ws.ctrl_f, 1, False,
                            ws.vstats[e, L.VS_GRADNORM:L.VS_GRADNORM + 1])
            if pol_u != val:         # (an unpaired step -- epoch_policy != epoch_baseline -- repacks separately)
                K.epoch_pack([(m.actor, ws.pk_actor)] if pol_u else [(m.critic, ws.pk_critic)])
        if not ws.merged_tail:
            K.value_finalize(ws.vpart, Ev, ws.vpart.shape[1], ws.vstats, L.VS_STRIDE)

    def _enqueue_tail_exchange(self, ws, obs0, actions0, behave0):
        """end of a data-parallel lock-step learn: ONE all-gather carries the final policy pass's
        loss sums, the value-loss moments of every epoch, the return moments and the z-filter
        column sums (four collectives otherwise); every rank then reduces the gathered rows in rank
        order, so the replicas stay bit-identical"""
        K, m, W = self.K, self.model, self.world_size
        Ep, Ev, D = self.epoch_policy, self.epoch_baseline, obs0.shape[1]
        mode = L.SMX_PPO_CLIP if self.ppo_mode == 'clip' else L.SMX_PPO_ADAPT
        c = ws.tail_cuts
        K.moments(ws.ret, ws.tp_ret)                           # _avg_return_targ (ppo.py:571)
        if self.use_z_filter:                                  # model.z_update(obs_iter)  (ppo.py:578-579)
            ws.zdelta.zero_()
            K.zfilter_update(obs0, ws.zdelta[:D], ws.zdelta[D:2 * D], ws.zdelta[2 * D:], obs0.shape[0])
        self._dist.all_gather_into_tensor(ws.tail_gather.view(-1), ws.tail_pack)
        G = ws.tail_gather
        torch.sum(G[:, :c[1]], 0, keepdim=True, out=ws.ppart_sum)
        K.policy_finalize(mode, ws.ppart_sum, 1, ws.g_surr, ws.g_kl, m.log_var.view(-1), ws.n_total,
                          ws.ctrl_f, Ep > 0, False, ws.dz3a,
                          ws.grads_a[m.actor.numel:m.actor.numel + self.action_dim],
                          ws.sumsq_a[ws.np_a:ws.np_a + 1], ws.pstats[Ep], dz3_t=ws.dz3aT)
        ws.vpart.view(Ev, W, ws.nblk_v, 8).copy_(G[:, c[1]:c[2]].reshape(W, Ev, ws.nblk_v, 8).permute(1, 0, 2, 3))
        K.value_finalize(ws.vpart, Ev, ws.vpart.shape[1], ws.vstats, L.VS_STRIDE)
        ws.mom_parts.copy_(G[:, c[2]:c[3]])
        K.moments_merge(ws.mom_parts, ws.ret_mom)
        if self.use_z_filter:
            torch.sum(G[:, c[3]:], 0, out=ws.zsum)
            m.z_filter.running_sum += ws.zsum[:D]
            m.z_filter.running_sumsq += ws.zsum[D:2 * D]
            m.z_filter.count += ws.zsum[2 * D:]

    def _enqueue_dp_epoch(self, ws, e, mode, aj, cj, actions0, behave0):
        """One lock-step epoch on several ranks with ONE collective.  The loss
        gradient is linear in dz3 = (g_surr + c_kl * g_kl) / n and only c_kl needs the global mean
        KL, so the backward pass runs on both right-hand sides (a third job in the same three
        launches) and the combination is formed after the all-reduce of
        [G_surr | G_critic | G_kl | loss partial rows]."""
        K, m = self.K, self.model
        adapt = mode == L.SMX_PPO_ADAPT
        self._clear_foreign_partials(ws)
        K.epoch_losses_dp(mode, ws.mean, m.log_var.view(-1), actions0, behave0, ws.ref_pol, ws.adv,
                          ws.ctrl_f, ws.dz3a, ws.dz3k if adapt else ws.g_kl, ws.ppart_ar, ws.n_total,
                          g_surr_t=ws.dz3aT, g_kl_t=ws.dz3kT if adapt else None, values=ws.vpred,
                          returns=ws.ret, v_dz3=ws.dz3c, v_partials=ws.vpart_loc_all[e])
        jobs = [aj]
        if adapt:
            jobs.append(dict(aj, dz3=ws.dz3k, dz2=ws.dz2k, dz1=ws.dz1k, grads=ws.grads_k,
                             sumsq=ws.sumsq_k, dz3T=ws.dz3kT, dz2T=ws.dz2kT, dz1T=ws.dz1kT))
        K.mlp3_backward_multi(jobs + [cj])
        self._dist.all_reduce(ws.ar)
        K.epoch_combine(mode, ws.ppart_ar, ws.ppart_ar.shape[0], ws.n_total, m.log_var.view(-1), ws.ctrl_f,
                        e > 0, True, ws.pstats[e], ws.grads_a, ws.grads_k if adapt else None,
                        m.actor.numel, ws.sumsq_a, ws.grads_c, ws.sumsq_c)
        K.clip_adam_pair((m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                          ws.sumsq_a, K.sumsq_blocks(ws.grads_a.numel()), True,
                          ws.pstats[e, L.PS_GRADNORM:L.PS_GRADNORM + 1]),
                         (m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                          ws.sumsq_c, K.sumsq_blocks(ws.grads_c.numel()), False,
                          ws.vstats[e, L.VS_GRADNORM:L.VS_GRADNORM + 1]),
                         ws.ctrl_f)

    def _enqueue_reward_filter(self, ws, rewards):
        """rewards * reward_scale, then RewardFilter.forward and .update (ppo.py:452-455,
        reward_filter.py:33-57): one launch into ws.rew.  Several ranks: the filter's statistics are those of
        the GLOBAL batch -- every rank normalises with the same state, the batch sums are all-reduced, and
        the state takes them in on every rank alike."""
        K = self.K
        if self.world_size == 1 or not self.use_r_filter:
            K.reward_filter(rewards, self.reward_scale, ws.rf_state, 1e-5, ws.rew, ws.rf_part, ws.rf_ticket,
                            use_filter=self.use_r_filter, update=self.use_r_filter)
            return ws.rew
        K.reward_filter(rewards, self.reward_scale, ws.rf_state, 1e-5, ws.rew, ws.rf_part, ws.rf_ticket,
                        use_filter=True, update=False, sums=ws.rf_sums)
        self._dist.all_reduce(ws.rf_sums)
        ws.rf_state[:2] += ws.rf_sums[:2]
        ws.rf_state[2:] = ws.rf_sums[2:]
        return ws.rew

    def _enqueue_optimize(self, ws, obs, obs_next, actions, rewards, dones, pds, pix=None,
                          pix_next=None):
        """the whole of _optimize (ppo.py:487-586) as a launch sequence"""
        # The reward statistics of several ranks are exchanged BEFORE everything else; the exchange's error word
        # (C_XCHG_ERR) is one of the control block's per-learn words, so those are zeroed in front of it -- zeroed
        # behind it (as the first launches of the step used to do) a timeout of this exchange would be erased.
        pre_zeroed = self.filter_rewards and self.use_r_filter and self.world_size > 1
        if pre_zeroed:
            ws.zero_block.zero_()
        if self.filter_rewards:
            rewards = self._enqueue_reward_filter(ws, rewards)
        if self.if_rnn_policy or self.model.if_pixel:
            return self._enqueue_optimize_stem(ws, obs, obs_next, actions, rewards, dones, pds, pix,
                                               pix_next, pre_zeroed)
        K, m, ref = self.K, self.model, self.ref_target_model
        B, N, D = obs.shape
        A = self.action_dim
        fused = ws.fused
        if not fused and not pre_zeroed:
            ws.zero_block.zero_()                # stop flag, epochs done, per-epoch policy statistics
        tail = self._enqueue_critic_pass(ws, obs, obs_next, filter_tail=not fused)

        obs0 = obs[:, 0, :]                      # ppo.py:527-537 (views, no copies)
        actions0 = actions[:, 0, :]
        behave0 = pds[:, 0, :]
        if fused:
            # ONE launch: both z-filtered copies of the step-0 observations (+ the transposed one), the
            # critic's obs_next rows, the reference policy's std columns, the packed weights of the
            # three networks, and the zeroing of the control block's per-learn words
            zf = m.z_filter if self.use_z_filter else None
            K.epoch_prepare(obs0, ws.xn, ws.xnT, ws.xr, zmean=zf._mean if zf else None, zstd=zf._std if zf else None,
                            ref_filter=ref.z_filter if self.use_z_filter else None,
                            obs_next=obs_next[:, 0, :] if tail is not None else None,
                            xnext=ws.xnext if tail is not None else None,
                            ref_log_var=ref.log_var.view(-1), ref_std=ws.ref_pol[:, A:],
                            pack=[(m.actor, ws.pk_actor), (m.critic, ws.pk_critic), (ref.actor, ws.pk_ref)],
                            zero_words=None if pre_zeroed else ws.zero_block.view(torch.int32))
        elif self.use_z_filter:
            zm, zs = m.z_filter._mean, m.z_filter._std          # refreshed by the critic pass
            K.zfilter_forward(obs0, zm, zs, ws.xn)
            rz = ref.z_filter               # statistics + filter in one launch
            K.zfilter_forward_sums(obs0, rz.running_sum, rz.running_sumsq, rz.count, rz.eps, ws.xr)
        else:
            ws.xn.copy_(obs0)
            ws.xr.copy_(obs0)
        # ref_pol = ref_target_model.forward_actor(obs_iter)   (ppo.py:539): the mean goes
        # straight into the left half of ref_pol, the std is exp(log_var) broadcast
        ref_job = dict(net=ref.actor, x=ws.xr, h1=ws.h1r, h2=ws.h2r, out=ws.ref_pol[:, :A],
                       act=L.SMX_ACT_TANH)
        if not fused:
            ws.xnT.copy_(ws.xn.t())
            ws.ref_pol[:, A:].copy_(torch.exp(ref.log_var).expand(ws.rows, A))
        if fused:
            self._enqueue_fused_epochs(ws, actions0, behave0, ref_job, tail,
                                       lambda: self._enqueue_gae_from_values(ws, obs, rewards, dones))
        else:
            # the reference policy and the critic's obs_next rows ride in epoch 0's forward
            # launches (four independent networks, one launch per layer); GAE follows them
            self._enqueue_lockstep_epochs(
                ws, actions0, behave0, first_extra=[ref_job] + ([tail] if tail is not None else []),
                after_first=lambda: self._enqueue_gae_from_values(ws, obs, rewards, dones))

        if ws.tail_deferred:
            self._enqueue_tail_exchange(ws, obs0, actions0, behave0)
            self._enqueue_final_stats(ws)
            return
        if fused and ws.merged_tail:
            # value-loss finalize, _avg_return_targ (ppo.py:571), model.z_update(obs_iter) (:578-579) and the
            # reported means: one launch
            K.learn_epilogue(ws.ret, ws.ret_mom, m.log_var.view(-1), ws.fin, ws.ticket[1:2],
                             zfilter=m.z_filter if self.use_z_filter else None, x=obs0, count_rows=B,
                             v_partials=ws.vpart, n_epochs=self.epoch_baseline, nblk=ws.vpart.shape[1],
                             v_stats=ws.vstats, stats_stride=L.VS_STRIDE)
            return
        K.moments(ws.ret, ws.ret_mom)           # _avg_return_targ (ppo.py:571)
        if self.world_size > 1:
            self._dist.all_gather_into_tensor(ws.mom_parts.view(-1), ws.ret_mom.clone())
            K.moments_merge(ws.mom_parts, ws.ret_mom)
        if self.use_z_filter:                   # model.z_update(obs_iter)  (ppo.py:578-579)
            if self.world_size > 1:
                ws.zdelta.zero_()
                K.zfilter_update(obs0, ws.zdelta[:D], ws.zdelta[D:2 * D], ws.zdelta[2 * D:], B)
                self._dist.all_reduce(ws.zdelta)
                m.z_filter.running_sum += ws.zdelta[:D]
                m.z_filter.running_sumsq += ws.zdelta[D:2 * D]
                m.z_filter.count += ws.zdelta[2 * D:]
            else:
                K.zfilter_update(obs0, m.z_filter.running_sum, m.z_filter.running_sumsq,
                                 m.z_filter.count, B)
        self._enqueue_final_stats(ws)

    # ======================================================================================
    # Policies with a shared stem: the LSTM stem (algo.rnn.if_rnn_policy, the reference
    # default) and / or the CNN stem (pixel observations).  A stem belongs to BOTH optimiser
    # groups (ppo_net.py:202-224), so the value epochs see the stem the policy epochs left
    # behind: the two loops run one after the other exactly as ppo.py:541-562 does (no
    # lock-step sharing here).  Stem input rows are [z(low_dim) | cnn(camera0 / 255)].
    # ======================================================================================
    def _stem_inputs(self, ws, mm, low2d, frames, xin, cnn_ws, zstats, stop=None):
        """xin[:, :D] = z-filter(low_dim rows) -- skipped when zstats is False (already there);
        xin[:, D:] = CNN features of `frames` under model `mm` (chunked over cnn_ws.F frames)"""
        K = self.K
        D = low2d.shape[1]
        if zstats is not False:
            if self.use_z_filter:
                K.zfilter_forward(low2d, zstats[0], zstats[1], ws.ztmp[:low2d.shape[0]])
                xin[:, :D].copy_(ws.ztmp[:low2d.shape[0]])
            else:
                xin[:, :D].copy_(low2d)
        if frames is not None:
            nF = frames.shape[0]
            # one chunk: the patch matrix of the first convolution survives until the frames change (the caller
            # bumps ws.frames_epoch when it refills them) -- every later forward of this learn, under either model,
            # and the backward passes reuse it
            tag = (getattr(ws, 'frames_epoch', 0), frames.data_ptr(), nF) if nF <= cnn_ws.F else None
            for f0 in range(0, nF, cnn_ws.F):
                f1 = min(nF, f0 + cnn_ws.F)
                mm._cnn_stem.forward(mm.cnn, frames[f0:f1], f1 - f0, cnn_ws, xin[f0:f1, D:], stop, cols1_tag=tag)

    def _enqueue_gae_stem(self, ws, obs, obs_next, pix, pix_next, rewards, dones):
        """critic over all (B, N+1) steps + windowed GAE with horizon H (ppo.py:376-418)"""
        K, m = self.K, self.model
        B, N, D = obs.shape
        rnn = self.if_rnn_policy
        ws.lcat[:, :N].copy_(obs)                       # torch.cat([obs, obs_next], 1)
        ws.lcat[:, N:].copy_(obs_next)
        frames = None
        if m.if_pixel:
            ws.fcat[:, :N].copy_(pix)
            ws.fcat[:, N:].copy_(pix_next)
            frames = ws.fcat.view((B * (N + 1),) + tuple(ws.fcat.shape[2:]))
        zst = m.z_filter.refresh_stats() if self.use_z_filter else None
        self._stem_inputs(ws, m, ws.lcat.view(B * (N + 1), D), frames, ws.xcat, ws.cnn_gae, zst)
        x = ws.xcat
        if rnn:
            x = self._lstm_forward_only(ws, m, ws.xcat, B, N + 1, ws.gatesG, ws.csG, ws.loG, ws.loG2)
        K.mlp3_forward(m.critic, x, ws.h1G, ws.h2G, ws.vals.view(-1, 1), L.SMX_ACT_NONE, pack=ws.pack_stem)
        H = self.horizon if rnn else N
        K.gae(ws.vals, rewards, dones, ws.gpow, ws.lpow, self.gamma, self.gamma ** H, B, N, H,
              ws.adv, ws.ret)
        if self.norm_adv:
            K.moments(ws.adv, ws.adv_mom)
            if self.world_size > 1:
                self._dist.all_gather_into_tensor(ws.mom_parts.view(-1), ws.adv_mom.clone())
                K.moments_merge(ws.mom_parts, ws.adv_mom)
            K.adv_normalize(ws.adv, ws.adv_mom, 1e-4)

    def _lstm_forward_only(self, ws, mm, xin, B, T, gates, cs, lo_a, lo_b):
        """the LSTM stack without anything kept for a backward pass (critic pass, reference policy):
        the layers share gates / cs scratch and ping-pong between two output buffers"""
        x, out, other = xin, lo_a, lo_b
        for layer, rnn in enumerate(mm.rnns):
            self.K.lstm_forward(rnn, x, B, T, ws.h0L[layer], ws.c0L[layer], gates, out, cs)
            x, out, other = out, other, out
        return x

    def _stem_forward(self, ws, mm, xin, stop, save=True):
        """features of the current stem weights -> the MLP input (ws.lo with the LSTM, xin without)"""
        K = self.K
        B, E = ws.key[0], ws.E
        if mm.if_pixel:
            self._stem_inputs(ws, mm, ws.low_it, ws.frames_it, xin, ws.cnn_it, False, stop)
        if not mm.if_rnn:
            return xin
        K.lstm_forward(mm.rnn, xin, B, E, ws.h0, ws.c0, ws.gates, ws.lo, ws.cs,
                       ws.hp if save else None, stop=stop)
        x = ws.lo
        for layer, up in enumerate(ws.upper, 1):       # stacked layers (rnn_layer > 1)
            K.lstm_forward(mm.rnns[layer], x, B, E, ws.h0L[layer], ws.c0L[layer], up.gates, up.lo, up.cs,
                           up.hp if save else None, stop=stop)
            x = up.lo
        return x

    def _stem_policy_forward(self, ws, e):
        K, m = self.K, self.model
        A, W = self.action_dim, self.world_size
        mode = L.SMX_PPO_CLIP if self.ppo_mode == 'clip' else L.SMX_PPO_ADAPT
        x = self._stem_forward(ws, m, ws.xn, ws.stop)
        K.mlp3_forward(m.actor, x, ws.h1a, ws.h2a, ws.mean, L.SMX_ACT_TANH, ws.stop, pack=ws.pack_stem)
        K.policy_loss(mode, ws.mean, m.log_var.view(-1), ws.act_it, ws.beh_it, ws.ref_pol, ws.adv,
                      ws.ctrl_f, ws.g_surr, ws.g_kl, ws.ppart)
        part, nblk = ws.ppart, ws.nblk_p
        if W > 1 and ws.stem_one_exchange and e < self.epoch_policy:
            # the loss sums ride on the gradient's all-reduce (_stem_policy_update finalises): here only what the backward
            # pass needs, which in clip mode is local -- dz3 = g_surr * (1 / n), the finalize's own expression with c_kl = 0
            torch.sum(ws.ppart, 0, keepdim=True, out=ws.ppart_sum)
            torch.mul(ws.g_surr, float(np.float32(1.0) / np.float32(ws.n_total)), out=ws.dz3a)
            return
        if W > 1:
            torch.sum(ws.ppart, 0, keepdim=True, out=ws.ppart_sum)
            self._dist.all_reduce(ws.ppart_sum)
            part, nblk = ws.ppart_sum, 1
        elif nblk > 4 * ws.ppart_fold.shape[0]:
            # thousands of partial rows (B * E ~ 10^5): folded to 64 first -- every workgroup of the finalize walks them all
            K.partials_fold(ws.ppart, nblk, ws.ppart_fold, ws.ctrl_f)
            part, nblk = ws.ppart_fold, ws.ppart_fold.shape[0]
        K.policy_finalize(mode, part, nblk, ws.g_surr, ws.g_kl, m.log_var.view(-1), ws.n_total,
                          ws.ctrl_f, e > 0, e < self.epoch_policy, ws.dz3a,
                          ws.grads_a[m.actor.numel:m.actor.numel + A],
                          ws.sumsq_a[ws.np_a:ws.np_a + 1], ws.pstats[e])

    def _stem_backward(self, ws, net, h1, h2, dz3, dz2, dz1, g_mlp, g_cnn, g_rnn, stop):
        """MLP backward, then back through the stems it sits on"""
        K, m = self.K, self.model
        B, E, D = ws.key[0], ws.E, ws.key[2]
        top = ws.upper[-1] if (m.if_rnn and ws.upper) else ws
        x = top.lo if m.if_rnn else ws.xn
        # (from FUSED_ROWS_MIN rows on: the three data-gradient products, d loss / d (LSTM output) among them, as one fused
        # launch -- kernels.mlp3_backward says whether it wrote dx)
        have_dx = K.mlp3_backward(net, x, h1, h2, dz3, dz2, dz1, g_mlp, None, stop, ws=ws.mlp_sk,
                                  packT=ws.packT_stem, dx=top.dlo if m.if_rnn else None)
        if m.if_rnn:
            # d loss / d (LSTM output) = dz1 . W1, then BPTT (dgates overwrite the saved gates)
            F = m.rnn_hidden
            if not have_dx:
                K.linear(dz1, 1, net.views['W1'], 0, None, top.dlo, ws.rows, F, net.H1, stop=stop)
            for layer in range(len(ws.upper), 0, -1):      # stacked layers, top down (rnn_layer > 1)
                up = ws.upper[layer - 1]
                below = ws.upper[layer - 2] if layer > 1 else ws
                o = m.rnn_offsets[layer]
                K.lstm_backward(m.rnns[layer], below.lo, B, E, ws.c0L[layer], up.gates, up.cs, up.hp, up.dlo,
                                up.gates, g_rnn[o:o + m.rnn_counts[layer]], stop, ws=ws.lstm_sk)
                # d loss / d (this layer's input) = dgates . W_ih
                K.linear(up.gates, 1, m.rnns[layer].views['weight_ih'], 0, None, below.dlo, ws.rows, F, 4 * F,
                         stop=stop)
            K.lstm_backward(m.rnn, ws.xn, B, E, ws.c0, ws.gates, ws.cs, ws.hp, ws.dlo, ws.gates,
                            g_rnn[:m.rnn_counts[0]], stop, ws=ws.lstm_sk)
            up, upW, upK = ws.gates, m.rnn.views['weight_ih'], 4 * m.rnn_hidden
        else:
            up, upW, upK = dz1, net.views['W1'], net.H1
        if m.if_pixel:
            # d loss / d (CNN features) = upstream . W[:, D:], times the feature ReLU's mask
            K.linear(up, 1, upW[:, D:], 0, None, ws.dxn[:, D:], ws.rows, m.cnn_feature_dim, upK,
                     relu_mask=ws.xn[:, D:], stop=stop)
            m._cnn_stem.backward(m.cnn, ws.rows, ws.cnn_it, ws.dxn[:, D:], g_cnn, stop)

    def _stem_policy_update(self, ws, e):
        K, m = self.K, self.model
        n_mlp, n0 = m.actor.numel, m.n_actor_block
        self._stem_backward(ws, m.actor, ws.h1a, ws.h2a, ws.dz3a, ws.dz2a, ws.dz1a,
                            ws.grads_a[:n_mlp], ws.grads_a[n0:n0 + m.n_cnn],
                            ws.grads_a[n0 + m.n_cnn:], ws.stop)
        if self.world_size > 1 and ws.stem_one_exchange:
            # [gradient | loss sums] in one all-reduce; then the finalize on the GLOBAL sums: statistics, log_var's gradient
            # (into its slot of the gradient, which the exchange left holding garbage: zero it first -- every rank the same),
            # the early-exit flag and the step counter, once, in front of the optimiser launch
            A = self.action_dim
            mode = L.SMX_PPO_CLIP
            ws.grads_a[n_mlp:n0].zero_()
            self._dist.all_reduce(ws.ar_pol)
            K.policy_finalize(mode, ws.ppart_sum, 1, ws.g_surr, ws.g_kl, m.log_var.view(-1), ws.n_total,
                              ws.ctrl_f, e > 0, True, ws.dz3a, ws.grads_a[n_mlp:n_mlp + A],
                              ws.sumsq_a[ws.np_a:ws.np_a + 1], ws.pstats[e])
        elif self.world_size > 1:
            # ONE all-reduce for the MLP and the stem.  log_var's gradient (between them) was built from
            # all-reduced loss sums and is global already: exactly one copy may enter the sum
            if self.rank != 0:
                ws.grads_a[n_mlp:n0].zero_()
            self._dist.all_reduce(ws.grads_a)
        K.sumsq_partials(ws.grads_a, ws.sumsq_a)
        K.clip_adam(m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                    ws.sumsq_a, K.sumsq_blocks(ws.grads_a.numel()), ws.ctrl_f, 0, True,
                    ws.pstats[e, L.PS_GRADNORM:L.PS_GRADNORM + 1])

    def _stem_value_epoch(self, ws, e):
        K, m = self.K, self.model
        x = self._stem_forward(ws, m, ws.xn, None)
        K.mlp3_forward(m.critic, x, ws.h1c, ws.h2c, ws.vpred.view(-1, 1), L.SMX_ACT_NONE, None, pack=ws.pack_stem)
        n_total = ws.n_total
        # (several ranks: the moments only feed statistics -- gathered once after the last epoch)
        K.value_loss(ws.vpred, ws.ret, n_total, ws.dz3c,
                     ws.vpart_loc_all[e] if self.world_size > 1 else ws.vpart[e], ws.ctrl_f, True)
        self._stem_backward(ws, m.critic, ws.h1c, ws.h2c, ws.dz3c.view(-1, 1), ws.dz2c, ws.dz1c,
                            ws.grads_c[m.n_stem:], ws.grads_c[:m.n_cnn], ws.grads_c[m.n_cnn:m.n_stem],
                            None)
        if self.world_size > 1:
            self._dist.all_reduce(ws.grads_c)
        K.sumsq_partials(ws.grads_c, ws.sumsq_c)
        K.clip_adam(m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                    ws.sumsq_c, K.sumsq_blocks(ws.grads_c.numel()), ws.ctrl_f, 1, False,
                    ws.vstats[e, L.VS_GRADNORM:L.VS_GRADNORM + 1])

    def _enqueue_optimize_stem(self, ws, obs, obs_next, actions, rewards, dones, pds, pix, pix_next,
                               pre_zeroed=False):
        """_optimize with the LSTM and / or CNN stem (ppo.py:487-586)"""
        K, m, ref = self.K, self.model, self.ref_target_model
        B, N, D = obs.shape
        A, E = self.action_dim, ws.E
        if not pre_zeroed:
            ws.zero_block.zero_()
        self._enqueue_gae_stem(ws, obs, obs_next, pix, pix_next, rewards, dones)

        # obs_iter: the first E steps of every sub-trajectory (E = 1 without the LSTM; ppo.py:521-537)
        ws.low_it.view(B, E, D).copy_(obs[:, :E])
        ws.act_it.view(B, E, A).copy_(actions[:, :E])
        ws.beh_it.view(B, E, 2 * A).copy_(pds[:, :E])
        if m.if_pixel:
            ws.frames_it.view((B, E) + tuple(pix.shape[2:])).copy_(pix[:, :E])
            ws.frames_epoch = getattr(ws, 'frames_epoch', 0) + 1      # new frames: the cached patch matrix is stale
        zst = (m.z_filter._mean, m.z_filter._std) if self.use_z_filter else None   # refreshed in the GAE pass
        rzst = ref.z_filter.refresh_stats() if self.use_z_filter else None
        self._stem_inputs(ws, m, ws.low_it, None, ws.xn, None, zst)     # CNN part: every forward
        self._stem_inputs(ws, ref, ws.low_it, ws.frames_it if m.if_pixel else None, ws.xr,
                          ws.cnn_it if m.if_pixel else None, rzst)
        # ref_pol = ref_target_model.forward_actor(obs_iter, cells)   (ppo.py:539)
        x = ws.xr
        if ref.if_rnn:
            x = self._lstm_forward_only(ws, ref, ws.xr, B, E, ws.gates, ws.cs, ws.lo,
                                        ws.upper[0].lo if ws.upper else None)
        K.mlp3_forward(ref.actor, x, ws.h1r, ws.h2r, ws.ref_mean, L.SMX_ACT_TANH, None, pack=ws.pack_stem)
        ws.ref_pol[:, :A].copy_(ws.ref_mean)
        ws.ref_pol[:, A:].copy_(torch.exp(ref.log_var).expand(ws.rows, A))

        self._stem_policy_forward(ws, 0)
        for e in range(self.epoch_policy):
            self._stem_policy_update(ws, e)
            self._stem_policy_forward(ws, e + 1)
        for e in range(self.epoch_baseline):
            self._stem_value_epoch(ws, e)
        if self.world_size > 1:
            Ev, W = self.epoch_baseline, self.world_size
            self._dist.all_gather_into_tensor(ws.vgather.view(-1), ws.vpart_loc_all.view(-1))
            ws.vpart.view(Ev, W, ws.nblk_v, 8).copy_(ws.vgather.permute(1, 0, 2, 3))
        K.value_finalize(ws.vpart, self.epoch_baseline, ws.vpart.shape[1], ws.vstats, L.VS_STRIDE)

        K.moments(ws.ret, ws.ret_mom)           # _avg_return_targ (ppo.py:571)
        if self.world_size > 1:
            self._dist.all_gather_into_tensor(ws.mom_parts.view(-1), ws.ret_mom.clone())
            K.moments_merge(ws.mom_parts, ws.ret_mom)
        if self.use_z_filter:                   # model.z_update(obs_iter)  (ppo.py:578-579)
            if self.world_size > 1:
                ws.zdelta.zero_()
                K.zfilter_update(ws.low_it, ws.zdelta[:D], ws.zdelta[D:2 * D], ws.zdelta[2 * D:], B * E, ws=ws.zscratch)
                self._dist.all_reduce(ws.zdelta)
                m.z_filter.running_sum += ws.zdelta[:D]
                m.z_filter.running_sumsq += ws.zdelta[D:2 * D]
                m.z_filter.count += ws.zdelta[2 * D:]
            else:
                K.zfilter_update(ws.low_it, m.z_filter.running_sum, m.z_filter.running_sumsq,
                                 m.z_filter.count, B * E, ws=ws.zscratch)
        self._enqueue_final_stats(ws)

    def _enqueue_final_stats(self, ws):
        self.K.final_stats(self.model.log_var.view(-1),
                           self.model.z_filter if self.use_z_filter else None, ws.fin)

    # ======================================================================================
    # _optimize / learn  (ppo.py:487-613)
    # ======================================================================================
    def _optimize(self, obs, actions, rewards, obs_next, persistent_infos, onetime_infos, dones):
        x = self._flat_obs(obs)
        xn = self._flat_obs(obs_next)
        pds = persistent_infos[-1].contiguous()
        actions, rewards, dones = actions.contiguous(), rewards.contiguous(), dones.contiguous()
        B, N, D = x.shape
        assert B == self.batch_size and N == self.n_step, \
            'batch is (%d, %d) but config says batch_size=%d n_step=%d' % (B, N, self.batch_size, self.n_step)
        pix = pix_next = None
        if self.model.if_pixel:                   # one camera (ppo_net.py:269 "assumes only one camera angle")
            pix = obs['pixel']['camera0'].contiguous()
            pix_next = obs_next['pixel']['camera0'].contiguous()
        ws = self._workspace(B, N, D, self.action_dim, pix.dtype if pix is not None else None)
        self._ensure_ctrl(ws)
        args = (x, xn, actions, rewards, dones, pds) + ((pix, pix_next) if pix is not None else ())
        if self.if_rnn_policy:
            # agent-side LSTM state at the head of every sub-trajectory (ppo.py:511-515):
            # (B, layers=1, H) -> the kernels' [B, H]
            nl, F, Fl = self.model.rnn_layers, self.model.rnn_hidden, self.model.rnn_hidden_logical
            if Fl != F:                 # (a hidden size padded to a multiple of 4: the pad of the state is zero)
                ws.h0L.zero_(); ws.c0L.zero_()
            ws.h0L[:, :, :Fl].copy_(onetime_infos[0].reshape(B, nl, Fl).transpose(0, 1))
            ws.c0L[:, :, :Fl].copy_(onetime_infos[1].reshape(B, nl, Fl).transpose(0, 1))
        if self.use_graph:
            # A captured graph is bound to the addresses of its inputs.  The first batch is captured
            # in place (a pointer-stable feed -- device-resident replay, the benchmark -- never pays
            # a copy).  As soon as a batch arrives somewhere else the learner switches, once, to
            # staging buffers of its own: every later batch is copied in (one pass over the batch,
            # ~0.1 ms at 226 MB) and the graph captured on the staging buffers is replayed --
            # instead of re-capturing ~130 launches for every new address.
            key = tuple(t.data_ptr() for t in args)
            g = self._graphs.get(key)
            if g is None and len(self._graphs) >= getattr(self, 'graph_input_sets', 1):
                if getattr(ws, 'staged', None) is None:
                    ws.staged = tuple(torch.empty_like(t) for t in args)
                for dst, src in zip(ws.staged, args):
                    dst.copy_(src)
                args = ws.staged
                key = tuple(t.data_ptr() for t in args)
                g = self._graphs.get(key)
            if g is None:
                # warm-up run outside capture (lazy module loads, hipFuncSetAttribute), on a
                # snapshot of the mutable state so that the captured replay is the first real step
                snap = self._snapshot_state()
                self._enqueue_optimize(ws, *args)
                torch.cuda.synchronize()
                self._restore_state(snap)
                g = torch.cuda.CUDAGraph() if self.world_size == 1 else _SegmentedGraph()
                # no cyclic garbage collection while the stream is capturing: a collection that
                # frees some earlier learner's graph or device tensors calls hipFree / graph
                # destructors in the middle of the capture, which aborts the process
                gc_was_enabled = gc.isenabled()
                gc.collect()
                gc.disable()
                c_cap = self._dist.count if self._dist is not None else 0
                try:
                    if self.world_size == 1:
                        with torch.cuda.graph(g):
                            self._enqueue_optimize(ws, *args)
                    else:
                        try:
                            g.capture(lambda: self._enqueue_optimize(ws, *args), self._dist)
                        except Exception as e:      # e.g. a runtime that refuses to capture next to RCCL
                            self.log.warning('graph segments unavailable (%r): eager launches from here on', e)
                            g = None
                            self.use_graph = False
                finally:
                    if gc_was_enabled:
                        gc.enable()
                self._restore_state(snap)
                if g is None:
                    self._enqueue_optimize(ws, *args)
                    return self._collect_stats(ws)
                # exchanges that run as kernels INSIDE the graph (PeerExchange) were counted while they were captured,
                # not executed: a replay re-runs them, so it re-counts them (the process group's calls count themselves)
                if self._dist is not None:
                    g.in_graph_collectives = self._dist.count - c_cap
                    self._dist.count = c_cap
                if getattr(ws, 'staged', None) is not None and args is ws.staged:
                    self._graphs = {key: g}          # staging mode: the in-place graph is retired
                else:
                    self._graphs[key] = g
            g.replay()
            if self._dist is not None:
                self._dist.count += getattr(g, 'in_graph_collectives', 0)
        else:
            self._enqueue_optimize(ws, *args)
        return self._collect_stats(ws)

    def _snapshot_state(self):
        m = self.model
        ws = self._ws
        st = [m.actor_flat, m.critic_flat, self.actor_exp_avg, self.actor_exp_avg_sq,
              self.critic_exp_avg, self.critic_exp_avg_sq, ws.scal]
        if self.use_z_filter:
            st += [m.z_filter.running_sum, m.z_filter.running_sumsq, m.z_filter.count]
        return [(t, t.clone()) for t in st]

    def _restore_state(self, snap):
        for t, c in snap:
            t.copy_(c)
        torch.cuda.synchronize()

    # what a learn() leaves behind for the host is read lazily (see _collect_stats)
    @property
    def trace(self):
        self._flush_stats()
        return self._trace

    @trace.setter
    def trace(self, v):
        self._trace = v

    @property
    def epochs_executed(self):
        self._flush_stats()
        return self._epochs_executed

    @epochs_executed.setter
    def epochs_executed(self, v):
        self._epochs_executed = v

    @property
    def kl_record(self):
        self._flush_stats()
        return self._kl_record

    @kl_record.setter
    def kl_record(self, v):
        self._kl_record = v

    def _collect_stats(self, ws):
        """The statistics of this learn(): ONE device->host read of the statistics block.  On a GPU
        the read is asynchronous (pinned buffer + event) and the returned mapping resolves itself
        when first looked at -- at the latest right after the NEXT learn() has been enqueued -- so
        the GPU never waits for the read-back and the Python that decodes it."""
        snap = {'beta': getattr(self, 'beta', None), 'clip_epsilon': getattr(self, 'clip_epsilon', None),
                'lr': self.actor_lr_scheduler.get_lr()[0],       # (each mode defines only its own)
                'n_sync': ws.n_sync}
        if not self.lazy_stats:
            return self._decode_stats(ws.scal.cpu(), snap)
        self._flush_stats()                        # the previous learn's, now that this one is queued
        if getattr(ws, 'scal_host', None) is None:
            ws.scal_host = torch.empty(ws.scal.shape, dtype=ws.scal.dtype, pin_memory=True)
        ws.scal_host.copy_(ws.scal, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        handle = DeferredStats(self._flush_stats)
        self._pending_stats = (ev, ws.scal_host, snap, handle)
        return handle

    def _flush_stats(self):
        """wait for the read-back in flight (if any) and decode it"""
        pend, self._pending_stats = self._pending_stats, None
        if pend is not None:
            ev, host, snap, handle = pend
            ev.synchronize()
            handle._value = self._decode_stats(host, snap)

    def _decode_stats(self, scal, snap):
        """host copy of the statistics block -> the reference's stats dict
        (ppo.py:219-224, 278-284, 328-331, 559-586)"""
        ctrl_i = scal[:L.CTRL_WORDS].view(torch.int32)
        o = L.CTRL_WORDS
        Ep, Ev = self.epoch_policy, self.epoch_baseline
        ps = scal[o:o + (Ep + 1) * L.PS_STRIDE].view(Ep + 1, L.PS_STRIDE).numpy(); o += (Ep + 1) * L.PS_STRIDE
        o += snap['n_sync']
        vs = scal[o:o + Ev * L.VS_STRIDE].view(Ev, L.VS_STRIDE).numpy(); o += Ev * L.VS_STRIDE
        ret_mom = scal[o + 3:o + 6].numpy()
        fin = scal[o + 8:o + 12].numpy()
        rf = scal[o + 12:o + 15].numpy()
        if int(ctrl_i[L.C_SYNC_ERR]) != 0:
            # the device is shared after all: from the next learn() on, the two-launch form (no in-launch wait).  The
            # parameters hold the state after the last COMPLETE epoch of the failed learn (every later Adam step of it
            # was skipped on the device); the caller decides whether to re-feed the batch.
            self._fb_timed_out = True
            self._ws = None
            self._graphs = {}
            raise RuntimeError('the in-launch wait of a fused forward + backward epoch timed out in the last learn() '
                               '(smx_epoch_fwdbwd_f32): a workgroup of the launch was not resident within 0.25 s -- the '
                               'device is shared with other kernels.  This learner now runs the two-launch epochs '
                               '(session_config.learner.exclusive_device = False selects them from the start)')
        if int(ctrl_i[L.C_XCHG_ERR]) != 0:
            raise RuntimeError('a peer exchange timed out in the last learn(): error word 0x%x (0x100 | phase << 4 | peer) '
                               '-- a rank died or fell behind by more than the timeout' % (int(ctrl_i[L.C_XCHG_ERR]) & 0xffff))
        done = int(ctrl_i[L.C_EPOCHS_DONE])      # policy updates applied
        # the reference breaks after the update whose KL is too large: `done` updates ran,
        # slot `done` holds the forward pass after the last one
        last = done - 1
        trace = {'policy': [], 'value': []}
        for e in range(done):
            d = {'_surr_loss': float(ps[e, L.PS_SURR]), '_entropy': float(ps[e, L.PS_ENTROPY]),
                 '_pol_kl': float(ps[e + 1, L.PS_KL])}
            if self.ppo_mode == 'clip':
                d['_clip_surr_loss'] = float(ps[e, L.PS_LOSS])
                d['_clip_epsilon'] = snap['clip_epsilon']
            else:
                d['_kl_loss_adapt'] = float(ps[e, L.PS_LOSS])
                d['_beta'] = snap['beta']
            if self.clip_actor_gradient:
                d['grad_norm_actor'] = float(ps[e, L.PS_GRADNORM])
            trace['policy'].append(d)
        for e in range(Ev):
            d = {'_val_loss': float(vs[e, L.VS_LOSS]), '_val_explained_var': float(vs[e, L.VS_EXPVAR])}
            if self.clip_critic_gradient:
                d['grad_norm_critic'] = float(vs[e, L.VS_GRADNORM])
            trace['value'].append(d)
        self._trace = trace
        self._epochs_executed = done
        stats = dict(trace['policy'][last])
        self._kl_record.append(stats['_pol_kl'])                         # ppo.py:559
        stats.update(trace['value'][-1])                                 # ppo.py:565-566
        stats['_avg_return_targ'] = float(ret_mom[1])
        stats['_avg_log_sig'] = float(fin[0])
        stats['_avg_behave_likelihood'] = float(ps[done, L.PS_LB])
        stats['_avg_is_weight'] = float(ps[done, L.PS_ISW])
        stats['_ref_behave_diff'] = float(ps[done, L.PS_REFBEH])
        stats['_lr'] = snap['lr']
        if self.use_z_filter:           # ppo.py:580-583, means formed on the device (final_stats)
            stats['obs_running_mean'] = float(fin[1])
            stats['obs_running_square'] = float(fin[2])
            stats['obs_running_std'] = float(fin[3])
        if self.use_r_filter:
            stats['reward_mean'] = float(rf[1] / rf[0])                  # reward_filter.py:59-63
        return stats

    def learn(self, batch):
        self.current_iteration += 1
        c0 = self._dist.count if self._dist is not None else 0
        batch = self._preprocess_batch_ppo(batch)
        tensorplex_update_dict = self._optimize(
            batch['obs'], batch['actions'], batch['rewards'], batch['obs_next'],
            batch['persistent_infos'], batch.get('onetime_infos'), batch['dones'])
        self.periodic_checkpoint(global_steps=self.current_iteration, score=None)
        self.tensorplex.add_scalars(tensorplex_update_dict, self.global_step)
        self.exp_counter += self._ws.B_total
        self.global_step += 1
        self.collectives_per_step = (self._dist.count - c0) if self._dist is not None else 0
        return tensorplex_update_dict

    def raw_values(self):
        """critic values of the last learn() as the reference lays them out: (B, N+1)"""
        ws = self._ws
        B, N = ws.key[0], ws.key[1]
        if ws.split_tail:
            return torch.cat([ws.vals[:B * N].view(B, N), ws.vals[B * N:].view(B, 1)], 1)
        return ws.vals.view(B, N + 1)

    # ---- reference-shaped accessors used by the parity tests ---------------------------------
    def _gae_and_return(self, obs, obs_next, rewards, dones):
        """-> (advantages (B,1), returns (B,1)) as ppo.py:355-418 returns them"""
        x, xn = self._flat_obs(obs), self._flat_obs(obs_next)
        B, N, D = x.shape
        ws = self._workspace(B, N, D, self.action_dim,
                             obs['pixel']['camera0'].dtype if self.model.if_pixel else None)
        self._ensure_ctrl(ws)
        if self.if_rnn_policy or self.model.if_pixel:
            if self.if_rnn_policy:
                if self.cells is None:
                    ws.h0.zero_(); ws.c0.zero_()
                else:
                    Fl = self.model.rnn_hidden_logical
                    ws.h0.zero_(); ws.c0.zero_()
                    ws.h0[:, :Fl].copy_(self.cells[0].reshape(B, -1)); ws.c0[:, :Fl].copy_(self.cells[1].reshape(B, -1))
            pix = obs['pixel']['camera0'] if self.model.if_pixel else None
            pixn = obs_next['pixel']['camera0'] if self.model.if_pixel else None
            self._enqueue_gae_stem(ws, x, xn, pix, pixn, rewards.contiguous(), dones.contiguous())
        else:
            self._enqueue_gae(ws, x, xn, rewards.contiguous(), dones.contiguous())
        return ws.adv.view(B, -1).clone(), ws.ret.view(B, -1).clone()

    # ======================================================================================
    # publish / schedule (ppo.py:615-682)
    # ======================================================================================
    def module_dict(self):
        return {'ppo': self.model}

    def publish_parameter(self, iteration, message=''):
        if self.exp_counter >= self.learner_config.parameter_publish.exp_interval:
            self._publish(iteration, message=message)
            self._post_publish()

    def _post_publish(self):
        final_kl = np.mean(self.kl_record)
        scale_c = self.learner_config.algo.clip_consts.scale_constant
        scale_a = self.learner_config.algo.adapt_consts.scale_constant
        if self.ppo_mode == 'clip':
            if final_kl > self.kl_target * self.clip_adjust_threshold[1]:
                if self.clip_lower < self.clip_epsilon:
                    self.clip_epsilon = self.clip_epsilon / scale_c
            elif final_kl < self.kl_target * self.clip_adjust_threshold[0]:
                if self.clip_upper > self.clip_epsilon:
                    self.clip_epsilon = self.clip_epsilon * scale_c
        else:
            if final_kl > self.kl_target * self.beta_adjust_threshold[1]:
                if self.beta_upper > self.beta:
                    self.beta = self.beta * scale_a
            elif final_kl < self.kl_target * self.beta_adjust_threshold[0]:
                if self.beta_lower < self.beta:
                    self.beta = self.beta / scale_a
        self.ref_target_model.update_target_params(self.model)
        self.kl_record = []
        self.exp_counter = 0
        self.actor_lr_scheduler.step()
        self.critic_lr_scheduler.step()

    def checkpoint_attributes(self):
        return ['model', 'ref_target_model', 'actor_lr_scheduler', 'critic_lr_scheduler',
                'current_iteration']

    def _prefetcher_preprocess(self, batch, out=None):
        return self.aggregator.aggregate(batch, out=out)
    _prefetcher_preprocess.accepts_out = True       # (LearnerDataPrefetcher: aggregate straight into pinned staging)
