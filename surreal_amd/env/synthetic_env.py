"""
Synthetic environments.  The reference has no synthetic env (it steps MuJoCo / Robosuite
simulators, one OS process per agent: surreal/agent/base.py:244-271, surreal/env/make_env.py);
BASELINE.json's configs[4] is a synthetic 1024-actor workload, so the dynamics are defined here
once and implemented twice with bit-identical fp32 results:

  * ``SyntheticEnv``    -- one actor, numpy, the reference ``Env`` protocol (reset/step, nested
    obs dicts): drives the reference-style Agent + windowing wrappers in tests;
  * ``SyntheticVecEnv`` -- all actors of a GPU stepped by ONE HIP launch
    (smx_synth_env_step_f32, csrc/smx_replay.hip) that also records the step straight into the
    device-resident rollout [actors, T, .]; ``emit_windows`` then cuts the rollout into
    n_step / stride sub-trajectories (smx_window_emit_f32) exactly as
    ExpSenderWrapperMultiStepMovingWindowWithInfo would have, without the data ever leaving HBM.

Dynamics (all fp32, no fused multiply-add):
    a      = clip(action, -1, 1)
    s'[k]  = clamp(0.9*s[k] + 0.5*a[k % A] + 0.01*((37k) % 17 - 8), -10, 10)
    reward = -0.1 * sum_j a[j]^2 + 0.05 * s'[0]      (accumulated in fp64, rounded once)
    done   = (t + 1 >= episode_len)
"""
import collections

import numpy as np
import torch

from surreal_amd import kernels as KN
from surreal_amd import _lib as L
from .base import Env


def _drift(D):
    k = np.arange(D)
    return (np.float32(0.01) * ((37 * k) % 17 - 8).astype(np.float32)).astype(np.float32)


class SyntheticEnv(Env):
    def __init__(self, obs_dim, action_dim, episode_len=200, seed=0, pixel=None):
        """pixel = (C, H, W): also emit a uint8 camera frame obs['pixel']['camera0'] (a fixed
        pattern shifted by the step count and the first state component)"""
        self.D, self.A, self.episode_len = obs_dim, action_dim, episode_len
        self.pixel = tuple(pixel) if pixel is not None else None
        self.rs = np.random.RandomState(seed)
        self.init_state = self.rs.randn(obs_dim).astype(np.float32)
        self.state = self.init_state.copy()
        self.t = 0

    def observation_spec(self):
        spec = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=(self.D,)))
        if self.pixel is not None:
            spec['pixel'] = collections.OrderedDict(camera0=self.pixel)
        return spec

    def _frame(self):
        C, H, W = self.pixel
        c, y, x = np.meshgrid(np.arange(C), np.arange(H), np.arange(W), indexing='ij')
        shift = 3 * self.t + int(100 * abs(float(self.state[0])))
        return ((37 * c + 5 * y + 11 * x + shift) % 256).astype(np.uint8)

    def action_spec(self):
        return {'dim': (self.A,), 'type': 'continuous'}

    def _obs(self):
        obs = collections.OrderedDict(
            low_dim=collections.OrderedDict(flat_inputs=self.state.copy()))
        if self.pixel is not None:
            obs['pixel'] = collections.OrderedDict(camera0=self._frame())
        return obs

    def _reset(self):
        self.state = self.init_state.copy()
        self.t = 0
        return self._obs(), {}

    def _step(self, action):
        a = np.clip(np.asarray(action, dtype=np.float32).reshape(-1), -1.0, 1.0).astype(np.float32)
        k = np.arange(self.D)
        sn = (np.float32(0.9) * self.state + np.float32(0.5) * a[k % self.A]).astype(np.float32)
        sn = np.clip((sn + _drift(self.D)).astype(np.float32), -10.0, 10.0).astype(np.float32)
        reward = float(np.float32(-0.1 * np.sum(a.astype(np.float64) ** 2) + 0.05 * float(sn[0])))
        done = (self.t + 1 >= self.episode_len)
        self.t += 1
        self.state = sn
        return self._obs(), reward, done, {}


class SyntheticVecEnv(object):
    """n actors on one GPU; rollouts of T steps recorded on the device"""

    def __init__(self, n_actors, obs_dim, action_dim, episode_len=200, seeds=None, device=None,
                 kernels=None, pixel=None, frame_stacks=1):
        """pixel = (C, H, W): every actor also has a camera (SyntheticEnv's frame: a pattern shifted by the step
        count and the first state component), rendered on the device; frame_stacks = n: the policy observes the
        last n frames on the channel axis (FrameStackWrapper's rule, surreal/env/wrapper.py:407-472) -- the rollout
        stores ONE raw uint8 frame per step and the stacking is a gather (smx_frame_stack_u8)."""
        self.K = kernels or KN.default_kernels()
        self.device = device or KN.default_device()
        self.n, self.D, self.A, self.episode_len = n_actors, obs_dim, action_dim, episode_len
        self.pixel = tuple(pixel) if pixel is not None else None
        self.frame_stacks = int(frame_stacks) if pixel is not None else 1
        seeds = list(range(n_actors)) if seeds is None else list(seeds)
        init = np.stack([np.random.RandomState(s).randn(obs_dim).astype(np.float32) for s in seeds])
        self.init_state = torch.as_tensor(init).to(self.device)
        self.state = self.init_state.clone()
        self.t = 0
        self.rolls = None
        self.persistent = True        # rollout(): the one-launch kernel where the policy's shapes allow it

    def reset(self):
        self.state.copy_(self.init_state)
        self.t = 0
        return self.state

    def start_rollout(self, T, info_width=0):
        """allocate a device rollout of T steps.  Every roll has T + 1 rows per actor (the
        observation roll needs the observation after the last step; the others leave their last
        row unused) so that one env-step launch records all fields.  A rollout is one episode
        segment: it starts right after a reset and T <= episode_len, so `done` can only be set
        on the last recorded step and windows never straddle an episode boundary."""
        assert self.t == 0 and T <= self.episode_len, 'rollouts start at an episode boundary'
        f = lambda *s: torch.zeros(*s, device=self.device, dtype=torch.float32)  # noqa: E731
        self.T = T
        R = T + 1
        self.rolls = {'obs': f(self.n, R, self.D), 'actions': f(self.n, R, self.A),
                      'rewards': f(self.n, R), 'dones': f(self.n, R)}
        if info_width:
            self.rolls['pds'] = f(self.n, R, info_width)
        self.slot = 0
        if self.pixel is not None:
            # raw camera frames, one per step, uint8 [actors, T + 1, C, H, W]; row 0 = the frame after reset
            self.frames = torch.zeros((self.n, R) + self.pixel, device=self.device, dtype=torch.uint8)
            self.K.synth_frames(self.state[:, 0], self.t, self.frames[:, 0])

    def observation(self):
        """what the policies see now: the state [n, D], or with a camera the nested observation
        {'pixel': {'camera0': uint8 [n, frame_stacks * C, H, W]}, 'low_dim': {'flat_inputs': state}}"""
        if self.pixel is None:
            return self.state
        C, H, W = self.pixel
        stacked = torch.empty(self.n, self.frame_stacks * C, H, W, device=self.device, dtype=torch.uint8)
        self.K.frame_stack(self.frames, self.frame_stacks, self.slot, 1, 1, 1, stacked)
        return collections.OrderedDict(pixel=collections.OrderedDict(camera0=stacked),
                                       low_dim=collections.OrderedDict(flat_inputs=self.state))

    def step(self, actions, pds=None):
        """actions [n, A] on the device -> next observation [n, D] (the state tensor)"""
        r = self.rolls
        if r is not None and self.slot < self.T:
            self.K.synth_env_step(self.state, self.init_state, actions, self.t, self.episode_len,
                                  self.slot, r['obs'], r['actions'], r['rewards'], r['dones'])
            if pds is not None and pds.data_ptr() != r['pds'][:, self.slot].data_ptr():
                r['pds'][:, self.slot] = pds         # (an agent may have written the slot in place)
            self.slot += 1
            if self.pixel is not None:
                # the camera frame of the observation AFTER this step (the terminal one when the episode ends here):
                # rendered from the recorded next observation, at the step count it belongs to
                self.K.synth_frames(r['obs'][:, self.slot, 0], self.t + 1, self.frames[:, self.slot])
        else:
            self.K.synth_env_step(self.state, self.init_state, actions, self.t, self.episode_len, 0,
                                  None, None, None, None)
        self.t = 0 if self.t + 1 >= self.episode_len else self.t + 1
        return self.state

    def rollout(self, agent, eps=None):
        """A whole recorded rollout (start_rollout(T) first) under `agent`'s plain-MLP policy: ONE launch
        (smx_synth_rollout_f32: 16 actors per workgroup through all T steps) where the shapes allow it, else
        THREE launches per environment step (the two hidden layers, then one launch that forms the policy mean,
        samples the action, steps every actor, records the transition and z-filters the next observation).
        Same numbers (1e-6: the layers' fp32 summation order differs between the two) as
        ``for t: agent.act_batch(state) -> step(actions, pds)``.
        eps: [T, n, A] standard-normal draws (default: drawn here in one launch; None-eps agents in
        a deterministic mode ignore it)."""
        T, n, K = self.T, self.n, self.K
        assert self.slot == 0 and 'pds' in self.rolls, 'start_rollout(T, info_width=2 * A) first'
        deterministic = agent.agent_mode in ('eval_deterministic', 'eval_deterministic_local')
        if eps is None and not deterministic:
            eps = torch.randn(T, n, self.A, device=self.device)
        if agent.rnn_config.if_rnn_policy or agent.model.if_pixel:
            return self._rollout_stem(agent, None if deterministic else eps)
        noise = agent.batch_noise(n).view(-1)
        zf = agent.model.z_filter if agent.use_z_filter else None
        log_var = agent.model.log_var.view(-1)
        actor = agent.model.actor
        plain_mlp = not (agent.rnn_config.if_rnn_policy or agent.model.if_pixel)
        if plain_mlp and self.persistent and K.synth_rollout_supported(actor):
            # ONE launch for the whole rollout: a workgroup owns 16 actors and walks them through all T steps
            # (csrc/smx_rollout.hip).  The packed weight copy is refreshed here: the agent's parameters only change
            # between rollouts (fetch_parameter)
            if getattr(self, '_pk', None) is None or self._pk.numel() != K.epoch_packed_numel(actor):
                self._pk = torch.zeros(K.epoch_packed_numel(actor), device=self.device)
            K.epoch_pack([(actor, self._pk)])
            K.synth_rollout(actor, self._pk, L.SMX_ACT_TANH, self.state, self.init_state, log_var, noise,
                            None if deterministic else eps.contiguous(), self.t, self.episode_len, T, self.slot,
                            self.rolls, zf)
            self.slot += T
            for _ in range(T):
                self.t = 0 if self.t + 1 >= self.episode_len else self.t + 1
            return
        if getattr(self, '_xn', None) is None:
            self._xn = torch.empty(n, self.D, device=self.device)
        if zf is not None:
            K.zfilter_forward_sums(self.state, zf.running_sum, zf.running_sumsq, zf.count, zf.eps, self._xn)
        else:
            self._xn.copy_(self.state)
        if plain_mlp and actor.OUT <= 32:
            # THREE launches per environment step: the two hidden layers, then one launch that forms the policy
            # mean (output layer + tanh) per actor, samples, steps, records and z-filters the next observation
            if getattr(self, '_h1', None) is None or self._h1.shape != (n, actor.H1):
                self._h1 = torch.empty(n, actor.H1, device=self.device)
                self._h2 = torch.empty(n, actor.H2, device=self.device)
            v = actor.views
            for t in range(T):
                K.linear(self._xn, 1, v['W1'], 1, v['b1'], self._h1, n, actor.H1, actor.D, act=L.SMX_ACT_RELU)
                K.linear(self._h1, 1, v['W2'], 1, v['b2'], self._h2, n, actor.H2, actor.H1, act=L.SMX_ACT_RELU)
                K.synth_act_env_step_head(v['W3'], v['b3'], self._h2, L.SMX_ACT_TANH, self.state, self.init_state,
                                          log_var, noise, None if deterministic else eps[t], self.t,
                                          self.episode_len, self.slot, self.rolls, zf, self._xn)
                self.slot += 1
                self.t = 0 if self.t + 1 >= self.episode_len else self.t + 1
            return
        for t in range(T):                     # four launches per step: three policy layers + the step launch
            mean = agent.policy_mean(self._xn)
            K.synth_act_env_step(self.state, self.init_state, mean, log_var, noise,
                                 None if deterministic else eps[t], self.t, self.episode_len, self.slot,
                                 self.rolls, zf, self._xn)
            self.slot += 1
            self.t = 0 if self.t + 1 >= self.episode_len else self.t + 1

    def can_rollout_into(self, agent):
        """rollout_into() needs the one-launch kernel: a plain-MLP policy whose shapes it takes"""
        return (self.persistent and self.pixel is None and not agent.rnn_config.if_rnn_policy and not agent.model.if_pixel
                and self.K.synth_rollout_supported(agent.model.actor))

    def rollout_into(self, agent, out, eps=None):
        """A whole rollout recorded STRAIGHT INTO a replay's slots (Replay.reserve_batch(n, window_shapes(T)) ->
        `out`: obs [n, T, D], obs_next [n, 1, D], actions [n, T, A], rewards / dones [n, T], pds [n, T, 2A]): with
        stride == n_step == T the moving-window rule (exp_sender_wrapper.py:209-228) makes the one window of an actor
        its rollout, so nothing is cut and nothing is copied -- the one-launch kernel writes the fields where the
        learner will read them.  Starts at an episode boundary (reset() first), like start_rollout()."""
        K, n = self.K, self.n
        T = out['obs'].shape[1]
        assert self.t == 0 and T <= self.episode_len and self.can_rollout_into(agent)
        assert tuple(out['obs'].shape) == (n, T, self.D) and all(out[k].is_contiguous() for k in out)
        deterministic = agent.agent_mode in ('eval_deterministic', 'eval_deterministic_local')
        if eps is None and not deterministic:
            eps = torch.randn(T, n, self.A, device=self.device)
        actor = agent.model.actor
        if getattr(self, '_pk', None) is None or self._pk.numel() != K.epoch_packed_numel(actor):
            self._pk = torch.zeros(K.epoch_packed_numel(actor), device=self.device)
        K.epoch_pack([(actor, self._pk)])
        rolls = {'obs': out['obs'], 'actions': out['actions'], 'rewards': out['rewards'], 'dones': out['dones'],
                 'pds': out['pds'], 'obs_last': out['obs_next']}
        K.synth_rollout(actor, self._pk, L.SMX_ACT_TANH, self.state, self.init_state, agent.model.log_var.view(-1),
                        agent.batch_noise(n).view(-1), None if deterministic else eps.contiguous(), self.t,
                        self.episode_len, T, 0, rolls, agent.model.z_filter if agent.use_z_filter else None)
        for _ in range(T):
            self.t = 0 if self.t + 1 >= self.episode_len else self.t + 1

    def _rollout_stem(self, agent, eps):
        """policies with an LSTM and / or CNN stem: one batched act per step (PPOAgent.act_batch: the stem and the
        MLP for all actors at once) on the stacked observation, then the step launch; the LSTM state every actor
        held BEFORE each step is recorded (what the window that starts there carries as onetime_infos,
        ppo_agent.py:133-135)"""
        T, n = self.T, self.n
        rnn = agent.rnn_config.if_rnn_policy
        agent.reset_batch()                      # a rollout starts at an episode boundary: zero state
        if rnn and 'cells' not in self.rolls:
            nl, F = agent.rnn_config.rnn_layer, agent.rnn_config.rnn_hidden
            self.rolls['cells'] = torch.zeros(n, T + 1, 2, nl, F, device=self.device)
        for t in range(T):
            a, pd = agent.act_batch(self.observation(), eps=None if eps is None else eps[t],
                                    out_pd=self.rolls['pds'][:, self.slot])
            if rnn:
                h, c = agent.batch_cells_before                   # (layers, n, hidden) each
                self.rolls['cells'][:, self.slot, 0].copy_(h.permute(1, 0, 2))
                self.rolls['cells'][:, self.slot, 1].copy_(c.permute(1, 0, 2))
            self.step(a, pds=pd)

    def rollout_reference(self, agent, eps):
        """the same rollout with TWO launches per step out of the kernels the persistent one is built from: the
        row-block forward (smx_epoch_forward_f32: the means) and the head + step launch.  Bit-identical to
        ``rollout`` on the persistent kernel -- the parity test's reference; not used by the product loop."""
        T, n, K = self.T, self.n, self.K
        actor = agent.model.actor
        noise = agent.batch_noise(n).view(-1)
        zf = agent.model.z_filter if agent.use_z_filter else None
        xn = torch.empty(n, self.D, device=self.device)
        mean = torch.empty(n, actor.OUT, device=self.device)
        pk = torch.zeros(K.epoch_packed_numel(actor), device=self.device)
        K.epoch_pack([(actor, pk)])
        if zf is not None:
            K.zfilter_forward_sums(self.state, zf.running_sum, zf.running_sumsq, zf.count, zf.eps, xn)
        else:
            xn.copy_(self.state)
        ctrl = torch.zeros(L.CTRL_WORDS, device=self.device)
        for t in range(T):
            K.epoch_forward([dict(net=actor, packed=pk, x=xn, out=mean, act=L.SMX_ACT_TANH)], None, ctrl, n)
            K.synth_act_env_step(self.state, self.init_state, mean, agent.model.log_var.view(-1), noise,
                                 None if eps is None else eps[t], self.t, self.episode_len, self.slot, self.rolls, zf, xn)
            self.slot += 1
            self.t = 0 if self.t + 1 >= self.episode_len else self.t + 1

    def window_shapes(self, n_step):
        """per-experience shape of every field emit_windows produces (for Replay.reserve_batch)"""
        shp = {'obs': (n_step, self.D), 'obs_next': (1, self.D), 'actions': (n_step, self.A),
               'rewards': (n_step,), 'dones': (n_step,)}
        if self.rolls is not None and 'pds' in self.rolls:
            shp['pds'] = (n_step, self.rolls['pds'].shape[2])
        if self.pixel is not None:
            C, H, Wd = self.pixel
            shp['pixel'] = (n_step, self.frame_stacks * C, H, Wd)
            shp['pixel_next'] = (1, self.frame_stacks * C, H, Wd)
        return shp

    def emit_windows(self, n_step, stride, out=None):
        """-> dict of [n*W, n_step, .] sub-trajectories (+ obs_next [n*W, 1, D]) cut from the
        recorded rollout with the reference's moving-window rule (exp_sender_wrapper.py:209-228), the
        same index arithmetic as the host wrapper (env/exp_sender_wrapper.py): window w = steps
        [w * advance, w * advance + n_step), W = windows_per_episode(T, n_step, stride) per actor"""
        from surreal_amd.env.exp_sender_wrapper import window_advance, windows_per_episode
        T = self.T
        assert self.slot == T, 'rollout not complete'
        W = windows_per_episode(T, n_step, stride)
        stride = window_advance(n_step, stride)
        r, K, n = self.rolls, self.K, self.n
        f = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)  # noqa: E731
        if out is not None:
            # the caller's buffers (e.g. the replay table's next n*W rows: no copy on insert)
            out = dict(out)
            assert out['obs'].shape == (n * W, n_step, self.D) and all(t.is_contiguous() for t in out.values())
            out['rewards'] = out['rewards'].view(n * W, n_step, 1)
            out['dones'] = out['dones'].view(n * W, n_step, 1)
        else:
            out = {'obs': f(n * W, n_step, self.D), 'obs_next': f(n * W, 1, self.D),
                   'actions': f(n * W, n_step, self.A), 'rewards': f(n * W, n_step, 1),
                   'dones': f(n * W, n_step, 1)}
        K.window_emit(r['obs'], 0, n_step, stride, W, out['obs'])
        K.window_emit(r['obs'], n_step, 1, stride, W, out['obs_next'])
        K.window_emit(r['actions'], 0, n_step, stride, W, out['actions'])
        K.window_emit(r['rewards'].view(n, T + 1, 1), 0, n_step, stride, W, out['rewards'])
        K.window_emit(r['dones'].view(n, T + 1, 1), 0, n_step, stride, W, out['dones'])
        out['rewards'] = out['rewards'].view(n * W, n_step)
        out['dones'] = out['dones'].view(n * W, n_step)
        if 'pds' in r:
            if 'pds' not in out:
                out['pds'] = f(n * W, n_step, r['pds'].shape[2])
            K.window_emit(r['pds'], 0, n_step, stride, W, out['pds'])
        if self.pixel is not None:
            # the windows' camera observations: window cut and frame stacking in one gather over the raw frames
            C, H, Wd = self.pixel
            ns = self.frame_stacks
            u8 = lambda *s: torch.empty(*s, device=self.device, dtype=torch.uint8)  # noqa: E731
            if 'pixel' not in out:
                out['pixel'], out['pixel_next'] = u8(n * W, n_step, ns * C, H, Wd), u8(n * W, 1, ns * C, H, Wd)
            K.frame_stack(self.frames, ns, 0, n_step, stride, W, out['pixel'])
            K.frame_stack(self.frames, ns, n_step, 1, stride, W, out['pixel_next'])
        if 'cells' in r:
            # onetime_infos: the LSTM state at the FIRST step of every window (exp_sender_wrapper.py:236-242)
            cw = r['cells'][0, 0].numel()
            cells = f(n * W, 1, cw)
            K.window_emit(r['cells'].view(n, T + 1, cw), 0, 1, stride, W, cells)
            out['cells'] = cells.view((n * W,) + tuple(r['cells'].shape[2:]))
        return out

    def to_batch(self, f):
        """emit_windows' fields -> the learner's batch contract (MultistepAggregatorWithInfo, aggregator.py:106-262)"""
        obs = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=f['obs']))
        nxt = collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=f['obs_next']))
        if 'pixel' in f:
            obs['pixel'] = collections.OrderedDict(camera0=f['pixel'])
            nxt['pixel'] = collections.OrderedDict(camera0=f['pixel_next'])
        once = [f['cells'][:, 0], f['cells'][:, 1]] if 'cells' in f else None
        return {'obs': obs, 'obs_next': nxt, 'actions': f['actions'], 'rewards': f['rewards'], 'dones': f['dones'],
                'persistent_infos': [f['pds']], 'onetime_infos': once}
