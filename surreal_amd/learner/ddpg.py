"""
DDPGLearner for MI355X -- drop-in for ``surreal.learner.ddpg.DDPGLearner``
(surreal/learner/ddpg.py:12-440): same constructor, config keys, SSAR batch contract,
statistics keys; one iteration = target estimate, critic MSE step (Adam 1e-3), actor step
through the UPDATED critic (-mean Q, gradient value-clip 1, Adam 1e-4), hard / soft target
update (ddpg.py:244-352, 403-428), as a chain of HIP launches with no host synchronisation
until the statistics are read.

Dense layers run on the FP32-MFMA layer kernel (csrc/smx_gemm.hip); the critic's
"concatenate the action into layer 2" (builders.py:58-84) is done with row strides instead of a
copy: layer 1 writes into the first c1 columns of a [B, c1+A] buffer whose last A columns hold
the action.

The TD3 switches are built too (ddpg.py:119-147, 266-283, 312-319): ``use_double_critic`` (a
second critic with its own optimiser and target, y = min of the two targets, ``Q_policy2`` and
the second critic's loss reported as the reference does) and ``use_action_regularization``
(clipped noise on the target policy's action, drawn from numpy's global stream like the
reference's).  The iteration is captured once in a hipGraph and replayed: the Adam step count,
the learning rates and the hard-update decision live on the device.

Pixel observations (ddpg_net.py:37-88): the perception CNN (model/cnn_stem.py, the PPO stem's
kernels) runs on camera0 / 255, its features go in front of the low-dim vector, it is trained by
the critic loss (Adam with the critic's hyper-parameters) and follows the target updates; the actor
update reuses the features formed before the critic step, as the reference does.

use_layernorm (default off) runs layer by layer (_enqueue_iteration_ln), with every other switch (round 6: the double critic
and camera observations too); torchx's LayerNorm semantics are unpinned (its source is absent): taken as
torch.nn.LayerNorm over the features.
"""
import gc
import types

import numpy as np
import torch

from surreal_amd import _lib as L
from surreal_amd import kernels as KN
from surreal_amd.learner.aggregator import SSARAggregator, FrameStackPreprocessor
from surreal_amd.learner.base import Learner, DeferredStats
from surreal_amd.model.ddpg_net import DDPGModel
from surreal_amd.session import ConfigError


class DDPGLearner(Learner):
    def __init__(self, learner_config, env_config, session_config):
        super().__init__(learner_config, env_config, session_config)
        self.K = KN.default_kernels()
        self.device = KN.default_device()
        self.current_iteration = 0
        self.batch_size = self.learner_config.replay.batch_size
        self.discount_factor = self.learner_config.algo.gamma
        self.n_step = self.learner_config.algo.n_step
        self.is_pixel_input = self.env_config.get('pixel_input', False)
        self.use_layernorm = self.learner_config.model.use_layernorm
        net = self.learner_config.algo.network
        self.use_double_critic = net.use_double_critic
        self.use_action_regularization = net.use_action_regularization
        self._target_update_init()
        # data-parallel: every rank owns batch_size samples of the global batch (uniform replay shards
        # per GPU, ddpg_configs.py:89-93 / SURVEY.md 8(e)); gradients are averaged before each Adam step
        from surreal_amd.learner.ppo import _dist_info
        self._dist, self.world_size, self.rank = _dist_info()
        # several ranks: the iteration is ONE hipGraph too when its exchanges run as kernels over IPC-mapped peer buffers
        # (PeerExchange, set up and self-checked with the workspace); on the process group (RCCL) the launches stay eager
        self.use_graph = bool(self.session_config.learner.get('use_hip_graph', True)) and self.device != 'cpu'
        self.exchange_kind = None
        self._pending_stats = None
        self.lazy_stats = self.device != 'cpu' and bool(self.session_config.learner.get('lazy_stats', True))
        self.clip_actor_gradient = net.clip_actor_gradient
        self.actor_gradient_clip_value = net.actor_gradient_value_clip if self.clip_actor_gradient else 0.0
        self.clip_critic_gradient = net.clip_critic_gradient
        self.critic_gradient_clip_value = net.critic_gradient_value_clip if self.clip_critic_gradient else 0.0
        self.lr_actor, self.lr_critic = net.lr_actor, net.lr_critic
        self.actor_regularization = net.actor_regularization
        self.critic_regularization = net.critic_regularization
        self.action_dim = self.env_config.action_spec.dim[0]
        conv = self.learner_config.model.get('conv_spec', None) or {}
        mk = dict(obs_spec=self.env_config.obs_spec, action_dim=self.action_dim,
                  use_layernorm=self.use_layernorm,
                  actor_fc_hidden_sizes=self.learner_config.model.actor_fc_hidden_sizes,
                  critic_fc_hidden_sizes=self.learner_config.model.critic_fc_hidden_sizes,
                  conv_out_channels=conv.get('out_channels'), conv_kernel_sizes=conv.get('kernel_sizes'),
                  conv_strides=conv.get('strides'), conv_hidden_dim=conv.get('hidden_output_dim'),
                  device=self.device, kernels=self.K)
        self.model = DDPGModel(**mk)
        self.model_target = DDPGModel(**mk)
        self.model_target.load_state_dict(self.model.state_dict())       # hard_update (ddpg.py:175-176)
        z = torch.zeros_like
        self.is_pixel_input = self.model.is_pixel_input
        if self.is_pixel_input:         # the perception CNN is in the critic's optimiser (ddpg_net.py:57-61)
            self.perc_exp_avg, self.perc_exp_avg_sq = z(self.model.perception_flat), z(self.model.perception_flat)
        if self.use_double_critic:
            # TD3's second critic (ddpg.py:119-147, 162-166, 177-178): own parameters, optimiser, target
            self.model2 = DDPGModel(critic_only=True, **mk)
            self.model_target2 = DDPGModel(critic_only=True, **mk)
            self.model_target2.load_state_dict(self.model2.state_dict())
            self.critic2_exp_avg, self.critic2_exp_avg_sq = z(self.model2.critic_flat), z(self.model2.critic_flat)
            if self.is_pixel_input:
                self.perc2_exp_avg = z(self.model2.perception_flat)
                self.perc2_exp_avg_sq = z(self.model2.perception_flat)
        self.actor_exp_avg, self.actor_exp_avg_sq = z(self.model.actor_flat), z(self.model.actor_flat)
        self.critic_exp_avg, self.critic_exp_avg_sq = z(self.model.critic_flat), z(self.model.critic_flat)
        self.actor_step = 0
        self.critic_step = 0
        self.frame_stack_concatenate_on_env = self.env_config.get('frame_stack_concatenate_on_env', True)
        self.frame_stack_preprocess = FrameStackPreprocessor(self.env_config.get('frame_stacks', 1))
        self.aggregator = SSARAggregator(self.env_config.obs_spec, self.env_config.action_spec)
        self._ws = None
        # independent layers of an iteration share launches (_enqueue_iteration_levels); off: one launch per layer
        self.level_schedule = bool(self.session_config.learner.get('ddpg_level_schedule', True))
        # the iteration on row blocks (smx_ddpg_rows.hip): two chain launches instead of fifteen dense ones.  Unset: used
        # for batches of up to 1024 rows per rank -- 4-row workgroups, measured 0.134 against 0.190 ms per iteration at
        # batch 512; past that more rounds of 4-row workgroups stream the weights again and the level schedule is used.  True / False
        # force it on (where the shapes allow) / off.
        rs = self.session_config.learner.get('ddpg_row_schedule', None)
        self.row_schedule = None if rs is None else bool(rs)
        # ... with a group's weight gradients formed in its update launch (one rank)
        self.rows_fused_update = bool(self.session_config.learner.get('ddpg_rows_fused_update', True))

    # ---- target update (ddpg.py:389-428) ----------------------------------------------------
    def _target_update_init(self):
        cfg = self.learner_config.algo.network.target_update
        self.target_update_type = cfg.type
        if self.target_update_type == 'soft':
            self.target_update_tau = cfg.tau
        elif self.target_update_type == 'hard':
            self.target_update_counter = 0
            self.target_update_interval = cfg.interval
        else:
            raise ConfigError('Unsupported ddpg update type: {}'.format(cfg.type))

    # ---- batch (ddpg.py:186-242) ---------------------------------------------------------------
    def _to_dev(self, x):
        if torch.is_tensor(x):
            return x.to(self.device, torch.float32)
        return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device)

    def preprocess(self, batch):
        for key in ('obs', 'obs_next'):
            for modality in batch[key]:
                for k in batch[key][modality]:
                    v = batch[key][modality][k]
                    if modality == 'pixel':      # stay uint8: the patch kernel applies / 255 while reading
                        v = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
                        batch[key][modality][k] = v.to(self.device)
                    else:
                        batch[key][modality][k] = self._to_dev(v)
        for key in ('actions', 'rewards', 'dones'):
            batch[key] = self._to_dev(batch[key])
        return batch
    preprocess.device_move_only = True       # (start_prefetching: the pinned stager does exactly this move)

    def _workspace(self, B, D):
        if self._ws is not None and self._ws.key == (B, D):
            return self._ws
        m, A = self.model, self.action_dim
        f = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)  # noqa: E731
        ws = types.SimpleNamespace(key=(B, D))
        a, c1, c2 = m.actor, m.c1, m.c2
        ws.h1a, ws.h2a, ws.act = f(B, a.H1), f(B, a.H2), f(B, A)        # actor forward (s or s')
        ws.xcat, ws.h2c, ws.q = f(B, c1 + A), f(B, c2), f(B)            # critic forward
        ws.q_next, ws.y, ws.dz3 = f(B), f(B), f(B)
        ws.dz2, ws.dxcat = f(B, c2), f(B, c1 + A)
        ws.q_actor, ws.dz3a, ws.dz2a, ws.dz1a = f(B), f(B, A), f(B, a.H2), f(B, a.H1)
        ws.grads_c = torch.zeros_like(m.critic_flat)
        ws.grads_a = torch.zeros_like(m.actor_flat)
        ws.gc = {}
        o = 0
        for name, v in m.critic.items():
            ws.gc[name] = ws.grads_c[o:o + v.numel()].view(v.shape)
            o += v.numel()
        ws.stats = torch.zeros(8, device=self.device)
        # the Adam step count and the learning rates live on the device: one captured hipGraph of
        # the iteration is replayed while they change (smx_adam_step_dev_f32)
        ws.step = torch.full((1,), self.critic_step, dtype=torch.int32, device=self.device)
        ws.dev_step = self.critic_step
        ws.lr = torch.tensor([self.lr_actor, self.lr_critic], dtype=torch.float32, device=self.device)
        ws.lr_host = (self.lr_actor, self.lr_critic)
        ws.q_policy = f(B)
        # the levelled schedule (_enqueue_iteration_levels) keeps the four forward chains of an iteration apart: target
        # actor, target critic, critic, actor each have their own activations, so independent layers share a launch
        ws.h1a_t, ws.h2a_t = f(B, a.H1), f(B, a.H2)
        ws.xcat_t, ws.h2c_t = f(B, c1 + A), f(B, c2)
        ws.xcat_a, ws.h2c_a = f(B, c1 + A), f(B, c2)
        ws.dz3_actor = torch.full((B,), -1.0 / B, device=self.device)       # d(-mean Q) / dQ  (ddpg.py:331)
        # the batch is staged into fixed buffers (5 small copies) so that the graph's pointers hold
        A = self.action_dim
        ws.s_obs, ws.s_next, ws.s_act, ws.s_rew, ws.s_done = f(B, D), f(B, D), f(B, A), f(B), f(B)
        if self.use_action_regularization:
            ws.s_noise, ws.act_n = f(B, A), f(B, A)
        if self.use_double_critic:
            ws.xcat2, ws.h2c2, ws.q2, ws.q_next2 = f(B, c1 + A), f(B, c2), f(B), f(B)
            ws.y2, ws.dz3_2 = f(B), f(B)
            ws.grads_c2 = torch.zeros_like(self.model2.critic_flat)
            ws.gc2 = {}
            o = 0
            for name, v in self.model2.critic.items():
                ws.gc2[name] = ws.grads_c2[o:o + v.numel()].view(v.shape)
                o += v.numel()
            ws.stats2 = torch.zeros(8, device=self.device)
        if self.is_pixel_input:
            from surreal_amd.model.cnn_stem import CnnStem
            cnn = m.cnn
            Dx = m.input_dim
            ws.xf, ws.xnf, ws.dxin = f(B, Dx), f(B, Dx), f(B, Dx)
            ws.cnn = CnnStem.workspace(cnn, B, self.device, backward=True)
            ws.cnn.sk = m._cnn_stem.splitk_workspace(cnn, B, self.device)
            ws.cnn_t = CnnStem.workspace(cnn, B, self.device, backward=False)
            ws.grads_p = torch.zeros_like(m.perception_flat)
            if self.use_double_critic:
                ws.xf2, ws.xnf2 = f(B, Dx), f(B, Dx)
                ws.cnn2 = CnnStem.workspace(cnn, B, self.device, backward=True)
                ws.cnn2.sk = ws.cnn.sk
                ws.cnn_t2 = CnnStem.workspace(cnn, B, self.device, backward=False)
                ws.grads_p2 = torch.zeros_like(m.perception_flat)
            ws.s_pix = ws.s_pix_next = None              # staged frames (allocated in their dtype)
        if self.use_layernorm:
            ws.ln, ws.ln_t = m.ln_workspace(B, self.device), self.model_target.ln_workspace(B, self.device)
            if self.use_double_critic:
                ws.ln2, ws.ln_t2 = self.model2.ln_workspace(B, self.device), self.model_target2.ln_workspace(B, self.device)
                ws.xcat_t2 = f(B, c1 + A)
            ws.dn2, ws.dz1c = f(B, c2), f(B, c1)
            ws.dn2a, ws.dn1a = f(B, a.H2), f(B, a.H1)
            ws.ln_ws = f(max(self.K.layernorm_backward_ws_floats(B, k) for k in (c1, c2, a.H1, a.H2)))
            ws.ln_scr = f(2 * max(c1, c2))
            ws.ga = {}
            o = 0
            for name, v in list(m.actor.views.items()) + list(m.actor_ln.items()):
                ws.ga[name] = ws.grads_a[o:o + v.numel()].view(v.shape)
                o += v.numel()
        ws.graph = None
        self._rank_weight = 1.0
        ws.xerr = torch.zeros(1, dtype=torch.int32, device=self.device) if self.device != 'cpu' else None
        if self.world_size > 1:            # a rank's share of the global batch: the means are weighted sums
            mine = torch.tensor([B], dtype=torch.int64, device=self.device)
            every = torch.empty(self.world_size, dtype=torch.int64, device=self.device)
            self._dist.all_gather_into_tensor(every, mine)
            self._rank_weight = float(B) / float(sum(every.tolist()))
            self._setup_peer_exchange(ws)
        self._ws = ws
        return ws

    def _setup_peer_exchange(self, ws):
        """several ranks on one node: the gradient / statistics all-reduces of an iteration (ddpg.py:244-400 run on
        shards: SURVEY.md 8(e) "same pattern, 2 grad all-reduces per iteration") as kernels over IPC-mapped peer buffers
        -- set up and SELF-CHECKED once, collectively; any failure leaves the process group in place and the launches
        eager.  session_config.learner.peer_exchange = False keeps the process group."""
        d = self._dist
        d.err_word = ws.xerr
        want = bool(self.session_config.learner.get('peer_exchange', True)) and self.device != 'cpu'
        need = max(64, ws.grads_c.numel(), ws.grads_a.numel(),
                   ws.grads_p.numel() if self.is_pixel_input else 0,
                   ws.grads_c2.numel() if self.use_double_critic else 0)
        if want and (d.exchange is None or d.exchange.capacity < need):
            if d.exchange is not None:
                d._d.barrier()
                d.exchange.close()
                d.exchange = None
            from surreal_amd.distributed.peer_exchange import PeerExchange
            d.exchange = PeerExchange.create(d._d, need, timeout_s=float(self.session_config.learner.get(
                'peer_exchange_timeout_s', 5.0)))
        self.exchange_kind = 'peer buffers (%s)' % d.exchange.check_message if d.exchange is not None else 'process group'
        if d.exchange is None:
            self.use_graph = False            # process-group collectives are not captured

    def _critic_backward(self, ws, x, B, model=None, dz3=None, xcat=None, h2c=None, gc=None):
        """gradients of a critic's parameters from dz3 (dLoss/dQ; default: the first critic's
        buffers); leaves dLoss/d(xcat) in ws.dxcat (its last A columns are dLoss/d(action))"""
        K, m, A = self.K, model or self.model, self.action_dim
        xcat = ws.xcat if xcat is None else xcat
        h2c = ws.h2c if h2c is None else h2c
        gc = ws.gc if gc is None else gc
        c, c1, c2, D = m.critic, m.c1, m.c2, x.shape[1]
        dz3 = (ws.dz3 if dz3 is None else dz3).view(B, 1)
        K.linear(dz3, 1, c['W3'], 0, None, ws.dz2, B, c2, 1, relu_mask=h2c, lda=1, ldb=c2)
        # d/d(relu(layer1)) masked by relu', into the first c1 columns of dxcat
        K.linear(ws.dz2, 1, c['W2'], 0, None, ws.dxcat, B, c1, c2, relu_mask=xcat, ldb=c1 + A,
                 ldc=c1 + A)
        # d/d(action) into the last A columns (no mask)
        K.linear(ws.dz2, 1, c['W2'][:, c1:], 0, None, ws.dxcat[:, c1:], B, A, c2, ldb=c1 + A,
                 ldc=c1 + A)
        K.linear_wgrad(ws.dxcat, x, gc['W1'], gc['b1'], c1, D, B, ldz=c1 + A)
        K.linear_wgrad(ws.dz2, xcat, gc['W2'], gc['b2'], c2, c1 + A, B)
        K.linear_wgrad(dz3, h2c, gc['W3'], gc['b3'], 1, c2, B, ldz=1)

    def _perception_backward(self, ws, model, xin, cnn_ws, grads_p):
        """the critic loss's gradient through the perception CNN (ddpg.py:304-308: critic_loss.backward()
        reaches model.perception): d(critic layer 1 input) = dz1 . W1, masked by the feature ReLU,
        then the stem's own backward; ws.dxcat must still hold this critic's dz1"""
        K, A = self.K, self.action_dim
        c1, Dx = model.c1, model.input_dim
        K.linear(ws.dxcat, 1, model.critic['W1'], 0, None, ws.dxin, xin.shape[0], Dx, c1, relu_mask=xin,
                 lda=c1 + A, ldb=Dx)
        model._cnn_stem.backward(model.cnn, xin.shape[0], cnn_ws, ws.dxin[:, :model.feat_dim], grads_p)

    def _average_over_ranks(self, t):
        """a per-rank mean -> the mean over the global batch: weighted by the rank's share of it, then
        one all-reduce; a no-op for one rank"""
        if self.world_size > 1:
            t.mul_(self._rank_weight)
            self._dist.all_reduce(t)

    def _enqueue_iteration_levels(self, ws, x, xn, actions, rewards, done):
        """One DDPG iteration (ddpg.py:244-352; low-dimensional observations, one critic, one rank) scheduled by
        DEPENDENCY LEVEL: the layers of the target actor, the target critic, the critic and the actor that do not
        depend on each other share a launch (smx_linear_multi_f32), and so do a level's weight gradients --
        22 dependent launches where the layer-by-layer schedule takes ~40 (each ~6 us at batch 512: the
        iteration is launch-latency bound).  The arithmetic of every layer is the same kernel with the same
        operands as in _enqueue_iteration: identical results.
        The actor's forward pass for ITS update (ddpg.py:326-329) only reads the actor's parameters, which the
        critic update does not touch, so it rides in levels 1-3."""
        K, m, mt, A = self.K, self.model, self.model_target, self.action_dim
        B, D = x.shape
        a, ta, c, tc = m.actor.views, mt.actor.views, m.critic, mt.critic
        c1, c2, ld = m.c1, m.c2, m.c1 + A
        H1, H2 = m.actor.H1, m.actor.H2
        R, T = L.SMX_ACT_RELU, L.SMX_ACT_TANH
        # 1: first layers of all four chains (none depends on another) -- and the batch's actions into the last A columns
        # of the critic's concat buffer as a fifth problem of the launch, actions . I^T (exact for FINITE actions: one product
        # with 1, the rest with 0 -- the contract |a| <= 1 of ddpg.py:262-263, which the statistics launch checks on `actions`
        # itself; an Inf would turn its row's other columns into NaN here, -0.0 becomes +0.0); as a strided torch copy in
        # front it was a 9 us launch of its own
        if getattr(ws, 'eyeA', None) is None:
            ws.eyeA = torch.eye(A, device=self.device)
        K.linear_multi([('linear', xn, 1, ta['W1'], 1, ta['b1'], ws.h1a_t, B, H1, D, dict(act=R)),
                        ('linear', xn, 1, tc['W1'], 1, tc['b1'], ws.xcat_t, B, c1, D, dict(act=R, ldc=ld)),
                        ('linear', x, 1, c['W1'], 1, c['b1'], ws.xcat, B, c1, D, dict(act=R, ldc=ld)),
                        ('linear', x, 1, a['W1'], 1, a['b1'], ws.h1a, B, H1, D, dict(act=R)),
                        ('linear', actions, 1, ws.eyeA, 1, None, ws.xcat[:, c1:], B, A, A, dict(ldc=ld))])
        # 2
        K.linear_multi([('linear', ws.h1a_t, 1, ta['W2'], 1, ta['b2'], ws.h2a_t, B, H2, H1, dict(act=R)),
                        ('linear', ws.xcat, 1, c['W2'], 1, c['b2'], ws.h2c, B, c2, ld, dict(act=R)),
                        ('linear', ws.h1a, 1, a['W2'], 1, a['b2'], ws.h2a, B, H2, H1, dict(act=R))])
        # 3: the target policy's action lands in the target critic's concat buffer, the policy's own action in the
        # concat buffer of the actor update's critic pass
        K.linear_multi([('linear', ws.h2a_t, 1, ta['W3'], 1, ta['b3'], ws.xcat_t[:, c1:], B, A, H2, dict(act=T, ldc=ld)),
                        ('linear', ws.h2c, 1, c['W3'], 1, c['b3'], ws.q.view(B, 1), B, 1, c2, dict(act=0)),
                        ('linear', ws.h2a, 1, a['W3'], 1, a['b3'], ws.xcat_a[:, c1:], B, A, H2, dict(act=T, ldc=ld)),
                        ('linear', ws.h2a, 1, a['W3'], 1, a['b3'], ws.act, B, A, H2, dict(act=T))])   # (dense copy: tanh')
        # 4, 5: Q'(s', mu'(s'))
        K.linear(ws.xcat_t, 1, tc['W2'], 1, tc['b2'], ws.h2c_t, B, c2, ld, act=R)
        K.linear(ws.h2c_t, 1, tc['W3'], 1, tc['b3'], ws.q_next.view(B, 1), B, 1, c2, act=0)
        # 6: y, critic loss gradient, the iteration's Adam step count
        K.ddpg_critic_loss_step(ws.q, ws.q_next, rewards, done, pow(self.discount_factor, self.n_step), ws.y, ws.dz3,
                                ws.step)
        # 7-9: critic backward; a layer's weight gradient shares the launch of the next data gradient
        dz3 = ws.dz3.view(B, 1)
        K.linear_multi([('linear', dz3, 1, c['W3'], 0, None, ws.dz2, B, c2, 1, dict(relu_mask=ws.h2c, lda=1, ldb=c2)),
                        ('wgrad', dz3, ws.h2c, ws.gc['W3'], ws.gc['b3'], 1, c2, B, dict(ldz=1))])
        K.linear_multi([('linear', ws.dz2, 1, c['W2'], 0, None, ws.dxcat, B, c1, c2,
                         dict(relu_mask=ws.xcat, ldb=ld, ldc=ld)),
                        ('wgrad', ws.dz2, ws.xcat, ws.gc['W2'], ws.gc['b2'], c2, ld, B, {})])
        K.linear_wgrad(ws.dxcat, x, ws.gc['W1'], ws.gc['b1'], c1, D, B, ldz=ld)
        # 10
        K.adam_step_dev(m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                        ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
        # 11-13: Q(s, mu(s)) through the UPDATED critic; d(-mean Q)/d(layer 2) needs only that pass's ReLU mask
        K.linear(x, 1, c['W1'], 1, c['b1'], ws.xcat_a, B, c1, D, act=R, ldc=ld)
        K.linear(ws.xcat_a, 1, c['W2'], 1, c['b2'], ws.h2c_a, B, c2, ld, act=R)
        K.linear_multi([('linear', ws.h2c_a, 1, c['W3'], 1, c['b3'], ws.q_actor.view(B, 1), B, 1, c2, dict(act=0)),
                        ('linear', ws.dz3_actor.view(B, 1), 1, c['W3'], 0, None, ws.dz2, B, c2, 1,
                         dict(relu_mask=ws.h2c_a, lda=1, ldb=c2))])
        # 14, 15: d/d(action), through tanh
        K.linear(ws.dz2, 1, c['W2'][:, c1:], 0, None, ws.dz3a, B, A, c2, ldb=ld)
        K.tanh_backward(ws.dz3a, ws.act, ws.dz3a)
        # 16-18: actor backward (data gradients, weight gradients), 19: its Adam step
        K.mlp3_backward(m.actor, x, ws.h1a, ws.h2a, ws.dz3a, ws.dz2a, ws.dz1a, ws.grads_a, None)
        K.adam_step_dev(m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                        ws.lr[0:1], ws.step, self.actor_regularization, self.actor_gradient_clip_value)
        K.ddpg_stats(ws.q, ws.y, rewards, actions, ws.q_actor, ws.stats)
        for tgt, src in self._target_pairs(mt, m):
            if self.target_update_type == 'soft':
                K.soft_update(tgt, src, self.target_update_tau)
            else:
                K.hard_update_every(tgt, src, ws.step, self.target_update_interval)

    def _rows_dims(self, D, rows=None):
        """(D, A, H1, H2, c1, c2) when the row-block kernels take these shapes (for a batch of `rows`), else None"""
        m = self.model
        dims = (D, self.action_dim, m.actor.H1, m.actor.H2, m.c1, m.c2)
        return dims if self.K.ddpg_rows_supported(*dims, rows=rows) else None

    def _rows_args(self, ws, x, xn, actions, rewards, done):
        if getattr(ws, 'rows_args', None) is not None and ws.rows_key == (x.data_ptr(), xn.data_ptr(), actions.data_ptr(),
                                                                         rewards.data_ptr(), done.data_ptr()):
            return ws.rows_args
        K, m, mt = self.K, self.model, self.model_target
        dims = self._rows_dims(x.shape[1], x.shape[0])
        ws.rows_packed = torch.zeros(K.ddpg_rows_packed_floats(*dims), device=self.device)
        ws.rows_versions = None          # nothing packed yet
        if getattr(ws, 'stats_slots', None) is None:
            ws.stats_slots = torch.zeros(2, 8, pin_memory=torch.cuda.is_available())
        if not hasattr(ws, 'ga'):
            ws.ga, o = {}, 0
            for name, v in m.actor.views.items():
                ws.ga[name] = ws.grads_a[o:o + v.numel()].view(v.shape)
                o += v.numel()
        nets = {'actor': m.actor.views, 'critic': m.critic, 'target_actor': mt.actor.views, 'target_critic': mt.critic}
        io = dict(x=x, x_next=xn, actions=actions, rewards=rewards, dones=done, xcat=ws.xcat, h2c=ws.h2c, q=ws.q,
                  q_next=ws.q_next, y=ws.y, dz3=ws.dz3, dz2=ws.dz2, dxcat=ws.dxcat, h1a=ws.h1a, h2a=ws.h2a, act=ws.act,
                  q_actor=ws.q_actor, dz3a=ws.dz3a, dz2a=ws.dz2a, dz1a=ws.dz1a, step=ws.step)
        ws.rows_args = K.ddpg_rows_args(dims, nets, ws.rows_packed, io, pow(self.discount_factor, self.n_step))
        ws.rows_key = (x.data_ptr(), xn.data_ptr(), actions.data_ptr(), rewards.data_ptr(), done.data_ptr())
        return ws.rows_args

    def _enqueue_iteration_rows(self, ws, x, xn, actions, rewards, done):
        """One DDPG iteration (ddpg.py:244-352; low-dimensional observations, one critic) on ROW BLOCKS: a
        workgroup carries 4 batch rows through whole chains -- target actor -> target critic -> y, critic -> loss ->
        its data gradients, and the actor's forward pass in one launch; Q(s, mu(s)) through the updated critic -> the
        actor's data gradients in a second (smx_ddpg_rows.hip).  Weight gradients (sums over all rows), Adam, the
        target update and the statistics are the launches of the other schedules, on the same buffers: 10 launches where
        the level schedule takes 22.  The MFMA loop sums a layer's products in another order than smx_linear_f32:
        equal to the level schedule within fp32 rounding (tests/test_gpu_ddpg.py), not bit for bit."""
        K, m, mt, A = self.K, self.model, self.model_target, self.action_dim
        B, D = x.shape
        c1, c2, ld = m.c1, m.c2, m.c1 + A
        H1, H2 = m.actor.H1, m.actor.H2
        args = self._rows_args(ws, x, xn, actions, rewards, done)
        gc, ga = ws.gc, ws.ga
        # every network's weights in fragment order: the update launches below keep the copies current element by
        # element; a full pack only when something ELSE wrote parameters since (construction, a checkpoint, a fetched
        # state dict -- _rows_refresh, outside a captured graph)
        if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
            self._rows_refresh(ws)
        soft = self.target_update_type == 'soft'
        tgt = dict(tau=self.target_update_tau if soft else 0.0, interval=0 if soft else self.target_update_interval)
        # one rank: a group's weight gradients and its step are ONE launch (value clipping needs no norm over the group);
        # several: the gradients are averaged over the ranks between them
        fuse = self.world_size == 1 and self.rows_fused_update
        K.ddpg_rows_critic(args)
        if not fuse:
            K.linear_multi([('wgrad', ws.dxcat, x, gc['W1'], gc['b1'], c1, D, B, dict(ldz=ld)),
                            ('wgrad', ws.dz2, ws.xcat, gc['W2'], gc['b2'], c2, ld, B, {}),
                            ('wgrad', ws.dz3.view(B, 1), ws.h2c, gc['W3'], gc['b3'], 1, c2, B, dict(ldz=1))])
            self._average_over_ranks(ws.grads_c)
        K.ddpg_rows_update(args, 'critic', m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                           ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value,
                           target=mt.critic_flat, wgrad=fuse, **tgt)
        K.ddpg_rows_actor(args)
        if not fuse:
            K.linear_multi([('wgrad', ws.dz1a, x, ga['W1'], ga['b1'], H1, D, B, {}),
                            ('wgrad', ws.dz2a, ws.h1a, ga['W2'], ga['b2'], H2, H1, B, {}),
                            ('wgrad', ws.dz3a, ws.h2a, ga['W3'], ga['b3'], A, H2, B, {})])
            self._average_over_ranks(ws.grads_a)
        K.ddpg_rows_update(args, 'actor', m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                           ws.lr[0:1], ws.step, self.actor_regularization, self.actor_gradient_clip_value,
                           target=mt.actor_flat, wgrad=fuse, stats=ws.stats if fuse else None,
                           stats_host=ws.stats_slots if fuse and self.lazy_stats else None, **tgt)
        ws.stats_zero_copy = bool(fuse and self.lazy_stats)
        if not fuse:                 # (fused: the statistics are one more workgroup of the actor's launch)
            K.ddpg_stats(ws.q, ws.y, rewards, actions, ws.q_actor, ws.stats)
            self._average_over_ranks(ws.stats[:6])
        ws.rows_versions = self._rows_versions()

    def _rows_versions(self):
        """torch's write counters of the four parameter buffers: the HIP launches do not move them (raw pointers), anything
        that writes parameters through torch does"""
        m, mt = self.model, self.model_target
        return tuple(int(t._version) for t in (m.actor_flat, m.critic_flat, mt.actor_flat, mt.critic_flat))

    def _rows_refresh(self, ws):
        """the row schedule's fragment-order copies follow the parameters through ddpg_rows_update only: repack all of
        them when the parameters were written from outside since the last iteration (or never packed)"""
        args = getattr(ws, 'rows_args', None)
        if args is not None and getattr(ws, 'rows_versions', None) != self._rows_versions():
            self.K.ddpg_rows_pack(args)
            ws.rows_versions = self._rows_versions()

    @staticmethod
    def _target_pairs(mt, m):
        """(target, source) buffers of the target-network update (ddpg.py:344-352): actor + critic as the one buffer they
        share when both models have it"""
        if getattr(mt, 'ac_flat', None) is not None and getattr(m, 'ac_flat', None) is not None:
            return ((mt.ac_flat, m.ac_flat),)
        return ((mt.actor_flat, m.actor_flat), (mt.critic_flat, m.critic_flat))

    def _critic_backward_ln(self, ws, model, lw, x, xcat, dz3, gc):
        """gradients of a LayerNorm critic's parameters from dz3 = dLoss/dQ [B] (builders.py:58-84 with use_layernorm,
        backwards); leaves dLoss/d(layer-1 pre-LayerNorm output) in ws.dz1c (the perception CNN's way in)"""
        K, A = self.K, self.action_dim
        B, D = x.shape
        c, c1, c2 = model.critic, model.c1, model.c2
        ld = c1 + A
        dz3 = dz3.view(B, 1)
        K.linear_wgrad(dz3, lw.c_n2, gc['W3'], gc['b3'], 1, c2, B, ldz=1)
        K.linear(dz3, 1, c['W3'], 0, None, ws.dn2, B, c2, 1, lda=1, ldb=c2)                 # d/d(LN2 output)
        K.layernorm_backward(ws.dn2, lw.c_a2, lw.cm2, lw.cr2, c['ln2.W'], ws.dz2, gc['ln2.W'], gc['ln2.b'], ws.ln_ws,
                             relu_mask=True)
        K.linear_wgrad(ws.dz2, xcat, gc['W2'], gc['b2'], c2, c1 + A, B)
        K.linear(ws.dz2, 1, c['W2'], 0, None, ws.dxcat, B, c1 + A, c2, ldb=ld, ldc=ld)        # d/d([LN1 output | action])
        K.layernorm_backward(ws.dxcat[:, :c1], lw.c_a1, lw.cm1, lw.cr1, c['ln1.W'], ws.dz1c, gc['ln1.W'], gc['ln1.b'],
                             ws.ln_ws, relu_mask=True)
        K.linear_wgrad(ws.dz1c, x, gc['W1'], gc['b1'], c1, D, B)

    def _perception_backward_ln(self, ws, model, xin, cnn_ws, grads_p):
        """the critic loss through the perception CNN in front of a LayerNorm critic: d(layer-1 input) = dz1c . W1 under the
        feature ReLU, then the stem's own backward (ws.dz1c still holds THIS critic's layer-1 gradient)"""
        K = self.K
        c1, Dx = model.c1, model.input_dim
        K.linear(ws.dz1c, 1, model.critic['W1'], 0, None, ws.dxin, xin.shape[0], Dx, c1, relu_mask=xin, ldb=Dx)
        model._cnn_stem.backward(model.cnn, xin.shape[0], cnn_ws, ws.dxin[:, :model.feat_dim], grads_p)

    def _enqueue_iteration_ln(self, ws, x, xn, actions, rewards, done, pix=None, pix_next=None):
        """one DDPG iteration (ddpg.py:244-352) with use_layernorm = True: every hidden ReLU is followed by a LayerNorm
        (builders.py:42-48, 65-75), so the networks run layer by layer (smx_linear_f32 + smx_layernorm_*_f32) and the
        LayerNorms' affine parameters are part of the optimiser groups.  With the TD3 switches (a second LayerNorm critic
        with its own target and optimiser, ddpg.py:119-147, 266-283, 312-319) and camera observations (the perception CNN
        in front of both networks, trained by the critic loss, ddpg_net.py:37-88) as in the plain schedule."""
        K, m, mt, A = self.K, self.model, self.model_target, self.action_dim
        B = x.shape[0]
        low, low_next = x, xn
        if self.is_pixel_input:
            mt.perception_into(pix_next, low_next, ws.cnn_t, ws.xnf)
            m.perception_into(pix, low, ws.cnn, ws.xf)
            x, xn = ws.xf, ws.xnf
        D = x.shape[1]
        lw, lt = ws.ln, ws.ln_t
        c, c1, c2, ld = m.critic, m.c1, m.c2, m.c1 + A
        a, av, aln = m.actor, m.actor.views, m.actor_ln
        gamma_n = pow(self.discount_factor, self.n_step)
        # ---- target: y = r + gamma^n * Q'(s', mu'(s')) * (1 - done) ----
        mt.actor_forward_ln(xn, lt, ws.act)
        mt.critic_forward_ln(xn, ws.act, lt, ws.xcat_t, ws.q_next)
        q_next = ws.q_next
        if self.use_double_critic:           # y = min of the two targets; the noise reaches only the second (ddpg.py:266-283)
            a2 = ws.act
            if self.use_action_regularization:
                torch.add(ws.act, ws.s_noise, out=ws.act_n)
                ws.act_n.clamp_(-1.0, 1.0)
                a2 = ws.act_n
            xn2 = xn
            if self.is_pixel_input:
                self.model_target2.perception_into(pix_next, low_next, ws.cnn_t2, ws.xnf2)
                xn2 = ws.xnf2
            self.model_target2.critic_forward_ln(xn2, a2, ws.ln_t2, ws.xcat_t2, ws.q_next2)
            torch.minimum(ws.q_next, ws.q_next2, out=ws.q_next2)
            q_next = ws.q_next2
        # ---- critic update(s) ----
        m.critic_forward_ln(x, actions, lw, ws.xcat, ws.q)
        x2 = x
        if self.use_double_critic:
            if self.is_pixel_input:
                self.model2.perception_into(pix, low, ws.cnn2, ws.xf2)
                x2 = ws.xf2
            self.model2.critic_forward_ln(x2, actions, ws.ln2, ws.xcat2, ws.q2)
        K.ddpg_critic_loss_step(ws.q, q_next, rewards, done, gamma_n, ws.y, ws.dz3, ws.step)
        self._critic_backward_ln(ws, m, lw, x, ws.xcat, ws.dz3, ws.gc)
        if self.is_pixel_input:
            self._perception_backward_ln(ws, m, x, ws.cnn, ws.grads_p)
            self._average_over_ranks(ws.grads_p)
        self._average_over_ranks(ws.grads_c)
        K.adam_step_dev(m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                        ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
        if self.is_pixel_input:
            K.adam_step_dev(m.perception_flat, ws.grads_p, self.perc_exp_avg, self.perc_exp_avg_sq,
                            ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
        ws.q_policy.copy_(ws.q)
        if self.use_double_critic:           # ddpg.py:312-319
            m2 = self.model2
            K.ddpg_critic_loss(ws.q2, q_next, rewards, done, gamma_n, ws.y2, ws.dz3_2)
            self._critic_backward_ln(ws, m2, ws.ln2, x2, ws.xcat2, ws.dz3_2, ws.gc2)
            if self.is_pixel_input:
                self._perception_backward_ln(ws, m2, x2, ws.cnn2, ws.grads_p2)
                self._average_over_ranks(ws.grads_p2)
            self._average_over_ranks(ws.grads_c2)
            K.adam_step_dev(m2.critic_flat, ws.grads_c2, self.critic2_exp_avg, self.critic2_exp_avg_sq,
                            ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
            if self.is_pixel_input:
                K.adam_step_dev(m2.perception_flat, ws.grads_p2, self.perc2_exp_avg, self.perc2_exp_avg_sq,
                                ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
            K.ddpg_stats(ws.q2, ws.y2, rewards, actions, ws.q2, ws.stats2)    # (the SECOND critic's loss is what is reported)
            self._average_over_ranks(ws.stats2[:6])
        # ---- actor update through the UPDATED critic: loss = -mean Q(s, mu(s)); the features formed before the critic
        # step are reused (ddpg.py:287, 326-327: perception.detach()) ----
        dz3 = ws.dz3.view(B, 1)
        m.actor_forward_ln(x, lw, ws.act)
        m.critic_forward_ln(x, ws.act, lw, ws.xcat, ws.q_actor)
        K.fill(ws.dz3, -1.0 / B)
        K.linear(dz3, 1, c['W3'], 0, None, ws.dn2, B, c2, 1, lda=1, ldb=c2)
        K.layernorm_backward(ws.dn2, lw.c_a2, lw.cm2, lw.cr2, c['ln2.W'], ws.dz2, ws.ln_scr[:c2], ws.ln_scr[c2:2 * c2],
                             ws.ln_ws, relu_mask=True)                                        # (critic gradients discarded)
        K.linear(ws.dz2, 1, c['W2'][:, c1:], 0, None, ws.dxcat[:, c1:], B, A, c2, ldb=ld, ldc=ld)   # d/d(action)
        ws.dz3a.copy_(ws.dxcat[:, c1:])
        K.tanh_backward(ws.dz3a, ws.act, ws.dz3a)
        ga = ws.ga
        K.linear_wgrad(ws.dz3a, lw.n2, ga['W3'], ga['b3'], A, a.H2, B)
        K.linear(ws.dz3a, 1, av['W3'], 0, None, ws.dn2a, B, a.H2, A, ldb=a.H2)
        K.layernorm_backward(ws.dn2a, lw.a2, lw.am2, lw.ar2, aln['ln2.W'], ws.dz2a, ga['ln2.W'], ga['ln2.b'], ws.ln_ws,
                             relu_mask=True)
        K.linear_wgrad(ws.dz2a, lw.n1, ga['W2'], ga['b2'], a.H2, a.H1, B)
        K.linear(ws.dz2a, 1, av['W2'], 0, None, ws.dn1a, B, a.H1, a.H2, ldb=a.H1)
        K.layernorm_backward(ws.dn1a, lw.a1, lw.am1, lw.ar1, aln['ln1.W'], ws.dz1a, ga['ln1.W'], ga['ln1.b'], ws.ln_ws,
                             relu_mask=True)
        K.linear_wgrad(ws.dz1a, x, ga['W1'], ga['b1'], a.H1, D, B)
        self._average_over_ranks(ws.grads_a)
        K.adam_step_dev(m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                        ws.lr[0:1], ws.step, self.actor_regularization, self.actor_gradient_clip_value)
        K.ddpg_stats(ws.q_policy, ws.y, rewards, actions, ws.q_actor, ws.stats)
        self._average_over_ranks(ws.stats[:6])
        pairs = list(self._target_pairs(mt, m))
        if self.is_pixel_input:
            pairs.append((mt.perception_flat, m.perception_flat))
        if self.use_double_critic:
            pairs.append((self.model_target2.critic_flat, self.model2.critic_flat))
            if self.is_pixel_input:
                pairs.append((self.model_target2.perception_flat, self.model2.perception_flat))
        for tgt, src in pairs:
            if self.target_update_type == 'soft':
                K.soft_update(tgt, src, self.target_update_tau)
            else:
                K.hard_update_every(tgt, src, ws.step, self.target_update_interval)

    def _enqueue_iteration(self, ws, x, xn, actions, rewards, done, pix=None, pix_next=None):
        """one DDPG iteration (ddpg.py:244-352) as a launch sequence without host round trips"""
        if self.use_layernorm:
            return self._enqueue_iteration_ln(ws, x, xn, actions, rewards, done, pix, pix_next)
        if not (self.is_pixel_input or self.use_double_critic):
            rows = self.row_schedule if self.row_schedule is not None else x.shape[0] <= 1024
            if rows and self._rows_dims(x.shape[1], x.shape[0]) is not None:
                return self._enqueue_iteration_rows(ws, x, xn, actions, rewards, done)
            if self.level_schedule and self.world_size == 1:
                return self._enqueue_iteration_levels(ws, x, xn, actions, rewards, done)
        K, m, mt, A = self.K, self.model, self.model_target, self.action_dim
        B = x.shape[0]
        low, low_next = x, xn
        if self.is_pixel_input:
            # forward_perception (ddpg_net.py:67-78): [CNN(camera0 / 255) | low_dim]; the model's own
            # features are formed ONCE, before the critic update, and reused by the actor update
            # (ddpg.py:287, 326-327: perception.detach())
            mt.perception_into(pix_next, low_next, ws.cnn_t, ws.xnf)
            m.perception_into(pix, low, ws.cnn, ws.xf)
            x, xn = ws.xf, ws.xnf
        # ---- target: y = r + gamma^n * Q'(s', mu'(s')) * (1 - done) ----
        K.mlp3_forward(mt.actor, xn, ws.h1a, ws.h2a, ws.act, L.SMX_ACT_TANH)
        mt.critic_forward_into(xn, ws.act, ws.xcat, ws.h2c, ws.q_next)
        q_next, gamma_n = ws.q_next, pow(self.discount_factor, self.n_step)
        if self.use_double_critic:
            # TD3 (ddpg.py:266-283): y = min(y1, y2), the second target critic evaluated at the target
            # policy's action -- with clipped noise added when action regularisation is on (the
            # reference adds it AFTER the first critic's target was formed, so only y2 sees it).
            # r + t is monotone in t: min(r + t1, r + t2) = r + gamma^n (1 - d) min(Q1', Q2')
            a2 = ws.act
            if self.use_action_regularization:
                torch.add(ws.act, ws.s_noise, out=ws.act_n)
                ws.act_n.clamp_(-1.0, 1.0)
                a2 = ws.act_n
            xn2 = xn
            if self.is_pixel_input:                  # the second target has its own perception
                self.model_target2.perception_into(pix_next, low_next, ws.cnn_t2, ws.xnf2)
                xn2 = ws.xnf2
            self.model_target2.critic_forward_into(xn2, a2, ws.xcat2, ws.h2c2, ws.q_next2)
            torch.minimum(ws.q_next, ws.q_next2, out=ws.q_next2)
            q_next = ws.q_next2
        # ---- critic update(s) ----
        m.critic_forward_into(x, actions, ws.xcat, ws.h2c, ws.q)
        x2 = x
        if self.use_double_critic:
            if self.is_pixel_input:
                self.model2.perception_into(pix, low, ws.cnn2, ws.xf2)
                x2 = ws.xf2
            self.model2.critic_forward_into(x2, actions, ws.xcat2, ws.h2c2, ws.q2)
        K.ddpg_critic_loss_step(ws.q, q_next, rewards, done, gamma_n, ws.y, ws.dz3, ws.step)
        self._critic_backward(ws, x, B)
        if self.is_pixel_input:
            self._perception_backward(ws, m, x, ws.cnn, ws.grads_p)
            self._average_over_ranks(ws.grads_p)
        self._average_over_ranks(ws.grads_c)
        K.adam_step_dev(m.critic_flat, ws.grads_c, self.critic_exp_avg, self.critic_exp_avg_sq,
                        ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
        if self.is_pixel_input:
            K.adam_step_dev(m.perception_flat, ws.grads_p, self.perc_exp_avg, self.perc_exp_avg_sq,
                            ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
        ws.q_policy.copy_(ws.q)
        if self.use_double_critic:                       # ddpg.py:312-319
            m2 = self.model2
            K.ddpg_critic_loss(ws.q2, q_next, rewards, done, gamma_n, ws.y2, ws.dz3_2)
            self._critic_backward(ws, x2, B, model=m2, dz3=ws.dz3_2, xcat=ws.xcat2, h2c=ws.h2c2, gc=ws.gc2)
            if self.is_pixel_input:
                self._perception_backward(ws, m2, x2, ws.cnn2, ws.grads_p2)
                self._average_over_ranks(ws.grads_p2)
            self._average_over_ranks(ws.grads_c2)
            K.adam_step_dev(m2.critic_flat, ws.grads_c2, self.critic2_exp_avg, self.critic2_exp_avg_sq,
                            ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
            if self.is_pixel_input:
                K.adam_step_dev(m2.perception_flat, ws.grads_p2, self.perc2_exp_avg, self.perc2_exp_avg_sq,
                                ws.lr[1:2], ws.step, self.critic_regularization, self.critic_gradient_clip_value)
            # the reference reports the SECOND critic's loss as 'critic_loss' (it overwrites the
            # variable, ddpg.py:313) and adds Q_policy2
            K.ddpg_stats(ws.q2, ws.y2, rewards, actions, ws.q2, ws.stats2)
            self._average_over_ranks(ws.stats2[:6])
        # ---- actor update through the UPDATED critic: loss = -mean Q(s, mu(s)) ----
        K.mlp3_forward(m.actor, x, ws.h1a, ws.h2a, ws.act, L.SMX_ACT_TANH)
        m.critic_forward_into(x, ws.act, ws.xcat, ws.h2c, ws.q_actor)
        K.fill(ws.dz3, -1.0 / B)
        c, c1, c2 = m.critic, m.c1, m.c2
        K.linear(ws.dz3.view(B, 1), 1, c['W3'], 0, None, ws.dz2, B, c2, 1, relu_mask=ws.h2c, lda=1,
                 ldb=c2)
        K.linear(ws.dz2, 1, c['W2'][:, c1:], 0, None, ws.dxcat[:, c1:], B, A, c2, ldb=c1 + A,
                 ldc=c1 + A)
        ws.dz3a.copy_(ws.dxcat[:, c1:])        # dense [B, A]
        K.tanh_backward(ws.dz3a, ws.act, ws.dz3a)
        K.mlp3_backward(m.actor, x, ws.h1a, ws.h2a, ws.dz3a, ws.dz2a, ws.dz1a, ws.grads_a, None)
        self._average_over_ranks(ws.grads_a)
        K.adam_step_dev(m.actor_flat, ws.grads_a, self.actor_exp_avg, self.actor_exp_avg_sq,
                        ws.lr[0:1], ws.step, self.actor_regularization, self.actor_gradient_clip_value)
        K.ddpg_stats(ws.q_policy, ws.y, rewards, actions, ws.q_actor, ws.stats)
        self._average_over_ranks(ws.stats[:6])       # means over the global batch (max |a| stays local)
        # ---- target networks (ddpg.py:389-428) ----
        pairs = list(self._target_pairs(mt, m))
        if self.is_pixel_input:
            pairs.append((mt.perception_flat, m.perception_flat))
        if self.use_double_critic:
            pairs.append((self.model_target2.critic_flat, self.model2.critic_flat))
            if self.is_pixel_input:
                pairs.append((self.model_target2.perception_flat, self.model2.perception_flat))
        for tgt, src in pairs:
            if self.target_update_type == 'soft':
                K.soft_update(tgt, src, self.target_update_tau)
            else:
                K.hard_update_every(tgt, src, ws.step, self.target_update_interval)

    def _optimize(self, obs, actions, rewards, obs_next, done):       # ddpg.py:244-352
        x = obs['low_dim']['flat_inputs']
        xn = obs_next['low_dim']['flat_inputs']
        B, D = x.shape
        ws = self._workspace(B, D)
        if ws.dev_step != self.critic_step:          # restored from a checkpoint
            ws.step.fill_(self.critic_step)
            ws.rows_versions = None
        if ws.lr_host != (self.lr_actor, self.lr_critic):
            ws.lr_host = (self.lr_actor, self.lr_critic)
            ws.lr.copy_(torch.tensor(ws.lr_host, dtype=torch.float32))
        # (a batch sampled straight into staging_fields() is already where the captured iteration reads it)
        for dst, src in ((ws.s_obs, x), (ws.s_next, xn), (ws.s_act, actions.reshape(B, -1)), (ws.s_rew, rewards.reshape(-1)),
                         (ws.s_done, done.reshape(-1))):
            # "already staged" means the SAME buffer, not merely the same address: a strided / reshaped view that starts at
            # the staging buffer's address is copied like any other source
            if not (src.data_ptr() == dst.data_ptr() and src.shape == dst.shape and src.stride() == dst.stride()
                    and src.dtype == dst.dtype):
                dst.copy_(src)
        frames = ()
        if self.is_pixel_input:
            pix, pix_next = obs['pixel']['camera0'], obs_next['pixel']['camera0']
            if ws.s_pix is None or ws.s_pix.dtype != pix.dtype:
                ws.s_pix, ws.s_pix_next = torch.empty_like(pix), torch.empty_like(pix_next)
                ws.graph = None
            ws.s_pix.copy_(pix)
            ws.s_pix_next.copy_(pix_next)
            frames = (ws.s_pix, ws.s_pix_next)
        if self.use_action_regularization:
            # ddpg.py:268-274: policy_noise 0.2 clipped at 0.5, from numpy's global stream
            noise = np.clip(np.random.normal(0, 0.2, size=(self.batch_size, self.action_dim)), -0.5, 0.5)
            ws.s_noise.copy_(torch.as_tensor(noise, dtype=torch.float32))
        if self.use_graph and ws.graph is None:
            # capture after one eager iteration (lazy allocations, module load); its effects are
            # real: the capture itself executes nothing
            self._enqueue_iteration(ws, ws.s_obs, ws.s_next, ws.s_act, ws.s_rew, ws.s_done, *frames)
            gc.collect()
            gc.disable()               # a collection inside capture may free device memory: illegal
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue_iteration(ws, ws.s_obs, ws.s_next, ws.s_act, ws.s_rew, ws.s_done, *frames)
                ws.graph = g
            finally:
                gc.enable()
        elif ws.graph is not None:
            self._rows_refresh(ws)
            ws.graph.replay()
            if getattr(ws, 'rows_args', None) is not None:
                ws.rows_versions = self._rows_versions()
        else:
            self._enqueue_iteration(ws, ws.s_obs, ws.s_next, ws.s_act, ws.s_rew, ws.s_done, *frames)
        self.critic_step += 1
        self.actor_step += 1
        ws.dev_step = self.critic_step
        if self.target_update_type == 'hard':
            self.target_update_counter += 1
        return self._collect_stats(ws)

    def staging_fields(self, batch_size):
        """the buffers the captured iteration reads its batch from, by replay field name (low-dimensional observations):
        ``replay.sample_batch(B, out=learner.staging_fields(B))`` gathers the sample where learn() would otherwise copy
        it (five small copies per iteration, 6 % of one at batch 512)"""
        if self.is_pixel_input:
            raise NotImplementedError('staging_fields: low-dimensional observations')
        ws = self._workspace(int(batch_size), self.model.input_dim)
        return {'obs': ws.s_obs, 'obs_next': ws.s_next, 'actions': ws.s_act, 'rewards': ws.s_rew, 'dones': ws.s_done}

    def _collect_stats(self, ws):
        """the iteration's one read-back; asynchronous on a GPU (resolved when looked at, at the latest
        after the next iteration has been enqueued -- see learner/base.py DeferredStats)"""
        if not self.lazy_stats:
            return self._decode_stats(ws.stats.cpu(), ws.stats2.cpu() if self.use_double_critic else None)
        self._flush_stats()
        if getattr(ws, 'stats_host', None) is None:
            ws.stats_host = torch.empty(3, 8, pin_memory=True)
        if getattr(ws, 'stats_zero_copy', False):
            # the row schedule's last launch wrote the statistics into host-mapped memory itself (slot = the iteration's
            # Adam step & 1: the slot of the iteration before is still being read): no copy launch
            host0 = ws.stats_slots[self.critic_step & 1]
        else:
            ws.stats_host[0].copy_(ws.stats, non_blocking=True)
            host0 = ws.stats_host[0]
        if ws.xerr is not None and self.world_size > 1:      # a peer exchange that timed out in this iteration
            ws.stats_host[2, :1].view(torch.int32).copy_(ws.xerr, non_blocking=True)
        else:
            ws.stats_host[2].zero_()
        if self.use_double_critic:
            ws.stats_host[1].copy_(ws.stats2, non_blocking=True)
        # (two events in turn, recorded on a Stream object cached per raw handle: Event() + record() through
        # torch.cuda.current_stream() were 15 us of host time per iteration)
        raw = L.current_stream().value
        if getattr(ws, 'ev_raw', -1) != raw:
            ws.ev_raw, ws.ev_stream = raw, torch.cuda.current_stream()
            ws.ev_pair, ws.ev_turn = (torch.cuda.Event(), torch.cuda.Event()), 0
        ev = ws.ev_pair[ws.ev_turn]
        ws.ev_turn ^= 1
        ev.record(ws.ev_stream)
        handle = DeferredStats(self._flush_stats)
        self._pending_stats = (ev, ws.stats_host, handle, host0)
        return handle

    def _flush_stats(self):
        pend, self._pending_stats = self._pending_stats, None
        if pend is not None:
            ev, host, handle, host0 = pend
            ev.synchronize()
            if int(host[2, :1].view(torch.int32)[0]) != 0:
                raise RuntimeError('a peer exchange timed out in the last DDPG iteration: error word 0x%x (0x100 | phase << 4 '
                                   '| peer) -- a rank died or fell behind by more than the timeout'
                                   % (int(host[2, :1].view(torch.int32)[0]) & 0xffff))
            handle._value = self._decode_stats(host0, host[1] if self.use_double_critic else None)

    def _decode_stats(self, st, st2):
        st = st.numpy()
        amax = float(st[6])
        assert amax <= 1.0, 'actions must lie in [-1, 1] (ddpg.py:262-263), got |a| = %g' % amax
        out = {'actor_loss': float(st[0]), 'critic_loss': float(st[1]), 'action_norm': float(st[2]),
               'rewards': float(st[3]), 'Q_target': float(st[4]), 'Q_policy': float(st[5])}
        if st2 is not None:
            st2 = st2.numpy()
            out['critic_loss'], out['Q_policy2'] = float(st2[1]), float(st2[5])
        return out

    def learn(self, batch):
        self.current_iteration += 1
        batch = self.preprocess(batch)
        stats = self._optimize(batch['obs'], batch['actions'], batch['rewards'], batch['obs_next'],
                               batch['dones'])
        self.tensorplex.add_scalars(stats, global_step=self.current_iteration)
        self.periodic_checkpoint(global_steps=self.current_iteration, score=None)
        return stats

    def module_dict(self):
        return {'ddpg': self.model}

    def checkpoint_attributes(self):
        return ['current_iteration', 'model', 'model_target']

    def _prefetcher_preprocess(self, batch):
        if not self.frame_stack_concatenate_on_env:        # ddpg.py:430-440
            batch = self.frame_stack_preprocess.preprocess_list(batch)
        return self.aggregator.aggregate(batch)
