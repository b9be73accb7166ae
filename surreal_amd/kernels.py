"""
Tensor-level facade over the C ABI (include/surreal_amd.h): every method takes torch device
tensors (views allowed where a leading stride is part of the C signature), passes raw
pointers + sizes + the current HIP stream to libsurreal_amd.so, and returns nothing -- outputs
are written into caller-owned tensors, nothing allocates, nothing synchronises.

The learner / replay / agent classes talk to the GPU only through this object, which is what
lets the multi-process (gloo, CPU) tests exercise the sharding logic with a test double.
There is NO fallback in the product: `HipKernels()` raises when the library or a GPU is absent.
"""
import ctypes

import torch

from surreal_amd import _lib as L


def _row_stride(view, width):
    """leading stride (floats) of a 2-D view whose rows are `width` contiguous floats"""
    assert view.dim() == 2 and view.shape[1] == width and (width == 1 or view.stride(1) == 1), \
        (tuple(view.shape), view.stride())
    return view.stride(0) if view.shape[0] > 1 else max(view.stride(0), width)


class HipKernels(object):
    name = 'hip'

    def __init__(self):
        L.require_gpu()
        self.lib = L.load()

    @staticmethod
    def _st():
        return L.current_stream()

    # ---- z-filter -----------------------------------------------------------------------
    def zfilter_stats(self, rs, rsq, cnt, eps, mean_out, std_out):
        L.call('smx_zfilter_stats_f32', L.ptr(rs), L.ptr(rsq), L.ptr(cnt), rs.numel(), float(eps),
               L.ptr(mean_out), L.ptr(std_out), self._st())

    def zfilter_forward(self, x_view, mean, std, out):
        rows, D = x_view.shape
        L.call('smx_zfilter_forward_f32', L.ptr(x_view), _row_stride(x_view, D), rows, D,
               L.ptr(mean), L.ptr(std), L.ptr(out), self._st())

    def zfilter_forward_sums(self, x_view, rs, rsq, cnt, eps, out):
        """stats + forward in one launch (an acting agent's one observation per step)"""
        rows, D = x_view.shape
        L.call('smx_zfilter_forward_sums_f32', L.ptr(x_view), _row_stride(x_view, D), rows, D,
               L.ptr(rs), L.ptr(rsq), L.ptr(cnt), float(eps), L.ptr(out), self._st())

    def diaggauss_sample(self, mean, log_var, noise_scale, eps, actions, pd):
        """acting head: pd = [mean, exp(log_var) * noise], actions = clip(eps * std + mean, -1, 1);
        eps None: clip(mean).  actions / pd may be row-strided views (rollout slots)."""
        rows, A = mean.shape
        L.call('smx_diaggauss_sample_f32', L.ptr(mean), _row_stride(mean, A), L.ptr(log_var),
               L.ptr(noise_scale), L.ptr(eps), 0 if eps is None else _row_stride(eps, A), rows, A,
               L.ptr(actions), _row_stride(actions, A), L.ptr(pd),
               0 if pd is None else _row_stride(pd, 2 * A), self._st())

    def zfilter_update(self, x_view, rs, rsq, cnt, count_rows, ws=None):
        """ws: optional scratch of zfilter_update_ws_floats(rows, D) floats -- many rows are then summed by many
        workgroups (smx_zfilter_update_ws_f32)"""
        rows, D = x_view.shape
        if ws is not None:
            L.call('smx_zfilter_update_ws_f32', L.ptr(x_view), _row_stride(x_view, D), rows, D, L.ptr(rs),
                   L.ptr(rsq), L.ptr(cnt), float(count_rows), L.ptr(ws), ws.numel(), self._st())
            return
        L.call('smx_zfilter_update_f32', L.ptr(x_view), _row_stride(x_view, D), rows, D, L.ptr(rs),
               L.ptr(rsq), L.ptr(cnt), float(count_rows), self._st())

    def zfilter_update_ws_floats(self, rows, D):
        return int(self.lib.smx_zfilter_update_ws_floats(int(rows), int(D)))

    # ---- MLP ----------------------------------------------------------------------------
    def mlp3_packed_numel(self, net):
        nbytes = self.lib.smx_mlp3_packed_bytes(net.D, net.H1, net.H2, net.OUT)
        if nbytes == 0:
            raise L.SmxError('fused MLP kernel does not support D=%d H1=%d H2=%d OUT=%d'
                             % (net.D, net.H1, net.H2, net.OUT))
        return nbytes // 4

    def mlp3_pack(self, net, packed):
        L.call('smx_mlp3_pack_f32', ctypes.byref(net.desc), L.ptr(packed), packed.numel() * 4,
               self._st())

    def mlp3_pack_zstats(self, net, packed, zf):
        """mlp3_pack + the statistics of ZFilter `zf` (its _mean / _std buffers) in one launch"""
        L.call('smx_mlp3_pack_zstats_f32', ctypes.byref(net.desc), L.ptr(packed), packed.numel() * 4,
               L.ptr(zf.running_sum), L.ptr(zf.running_sumsq), L.ptr(zf.count), zf.running_sum.numel(),
               float(zf.eps), L.ptr(zf._mean), L.ptr(zf._std), self._st())

    def fused_exact_zfilter(self, on):
        """process-wide: the fused critic pass z-filters with the reference's division (z_filter.py:77) instead of
        (x - m) * (1 / s); the goldens run with the default (off), DESIGN.md 1"""
        L.call('smx_mlp3_fused_exact_zfilter', 1 if on else 0)

    def mlp3_forward_fused(self, packed, net, x_main, x_tail, zmean, zstd, out, act):
        G, T0, D = x_main.shape
        T1 = 0 if x_tail is None else x_tail.shape[1]
        assert x_main.is_contiguous() and (x_tail is None or x_tail.is_contiguous())
        L.call('smx_mlp3_forward_fused_f32', L.ptr(packed), net.D, net.H1, net.H2, net.OUT,
               L.ptr(x_main), L.ptr(x_tail), G, T0, T1, L.ptr(zmean), L.ptr(zstd), L.ptr(out),
               act, self._st())

    # rows from which the fused many-row forward pays: below, its 128-row workgroups leave most of the 256 CUs idle and
    # the layered GEMMs (32 x 32 / 64 x 64 tiles) win (7936 rows: 62 workgroups, measured equal)
    FUSED_ROWS_MIN = 24576

    def mlp3_forward(self, net, x, h1, h2, out, act, stop=None, pack=None):
        """forward keeping h1 / h2.  pack: scratch of mlp3_packed_numel(net) floats -- with it, a pass over >=
        FUSED_ROWS_MIN rows of a supported shape runs as ONE fused launch (smx_mlp3_forward_rows_f32)"""
        assert x.is_contiguous()
        if pack is not None and x.shape[0] >= self.FUSED_ROWS_MIN and h1.is_contiguous() and h2.is_contiguous() and \
                L.load().smx_mlp3_forward_rows_supported(net.D, net.H1, net.H2, net.OUT):
            ld = 0 if out.dim() < 2 or out.stride(0) == out.shape[1] else out.stride(0)
            rc = L.load().smx_mlp3_forward_rows_f32(ctypes.byref(net.desc), L.ptr(x), x.shape[0], L.ptr(h1), L.ptr(h2),
                                                    L.ptr(out), act, ld, L.ptr(pack), pack.numel() * 4, L.ptr(stop), self._st())
            if rc != L.SMX_E_UNSUPPORTED:
                L.check(rc, 'smx_mlp3_forward_rows_f32')
                return
        L.call('smx_mlp3_forward_f32', ctypes.byref(net.desc), L.ptr(x), x.shape[0], L.ptr(h1),
               L.ptr(h2), L.ptr(out), act, L.ptr(stop), self._st())

    def _jobs(self, jobs):
        arr = (L.Mlp3Job * len(jobs))()
        for k, j in enumerate(jobs):
            g = lambda name: L.ptr(j.get(name)) if j.get(name) is not None else None  # noqa: E731
            arr[k].net = ctypes.pointer(j['net'].desc)
            arr[k].x = g('x')
            arr[k].rows = j['x'].shape[0]
            arr[k].h1, arr[k].h2, arr[k].out = g('h1'), g('h2'), g('out')
            arr[k].out_act = int(j.get('act', 0))
            out = j.get('out')
            arr[k].out_ld = 0 if out is None or out.dim() < 2 or out.stride(0) == out.shape[1] \
                else out.stride(0)
            arr[k].dz3, arr[k].dz2, arr[k].dz1 = g('dz3'), g('dz2'), g('dz1')
            arr[k].grads, arr[k].sumsq_partials = g('grads'), g('sumsq')
            arr[k].stop_flag = g('stop')
            arr[k].h1T, arr[k].h2T, arr[k].xT = g('h1T'), g('h2T'), g('xT')
            arr[k].dz3T, arr[k].dz2T, arr[k].dz1T = g('dz3T'), g('dz2T'), g('dz1T')
            tv = j.get('h1T')      # all transposed views of a job share one row stride
            arr[k].ldT = 0 if tv is None else (tv.stride(0) if tv.shape[0] > 1 else tv.shape[1])
        return arr

    def mlp3_forward_multi(self, jobs):
        """jobs: list of dicts(net, x, h1, h2, out, act[, stop]) -- one launch per layer for all"""
        L.call('smx_mlp3_forward_multi_f32', self._jobs(jobs), len(jobs), self._st())

    def mlp3_backward_multi(self, jobs):
        """jobs: list of dicts(net, x, h1, h2, dz3, dz2, dz1, grads, sumsq[, stop])"""
        L.call('smx_mlp3_backward_multi_f32', self._jobs(jobs), len(jobs), self._st())

    def mlp3_backward_partials(self, net):
        return self.lib.smx_mlp3_backward_partials(net.D, net.H1, net.H2, net.OUT)

    def mlp3_backward(self, net, x, h1, h2, dz3, dz2, dz1, grads, sumsq, stop=None, ws=None, packT=None, dx=None):
        """ws: optional split-K workspace (>= mlp3_backward_ws_floats(net, rows) floats) for many-row calls that do
        not need the sum-of-squares partials: the weight gradients' rows are cut into chunks.
        packT (scratch of mlp3_dgrad_rows_ws_floats(net) floats) [+ dx (rows, D): wanted gradient w.r.t. x]: from
        FUSED_ROWS_MIN rows on the three data-gradient products run as ONE fused launch (smx_mlp3_backward_rows_f32).
        -> True when dx was written (the caller then skips its own dz1 . W1 product)"""
        if ws is not None and sumsq is None and packT is not None and x.shape[0] >= self.FUSED_ROWS_MIN and \
                dz2.is_contiguous() and dz1.is_contiguous() and (dx is None or dx.is_contiguous()):
            rc = L.load().smx_mlp3_backward_rows_f32(
                ctypes.byref(net.desc), L.ptr(x), L.ptr(h1), L.ptr(h2), L.ptr(dz3), x.shape[0], L.ptr(dz2), L.ptr(dz1),
                L.ptr(dx), L.ptr(grads), L.ptr(ws), ws.numel(), L.ptr(packT), packT.numel(), L.ptr(stop), self._st())
            if rc != L.SMX_E_UNSUPPORTED:
                L.check(rc, 'smx_mlp3_backward_rows_f32')
                return dx is not None
        if ws is not None and sumsq is None:
            L.call('smx_mlp3_backward_splitk_f32', ctypes.byref(net.desc), L.ptr(x), L.ptr(h1), L.ptr(h2),
                   L.ptr(dz3), x.shape[0], L.ptr(dz2), L.ptr(dz1), L.ptr(grads), L.ptr(ws), ws.numel(),
                   L.ptr(stop), self._st())
            return False
        L.call('smx_mlp3_backward_f32', ctypes.byref(net.desc), L.ptr(x), L.ptr(h1), L.ptr(h2),
               L.ptr(dz3), x.shape[0], L.ptr(dz2), L.ptr(dz1), L.ptr(grads), L.ptr(sumsq),
               L.ptr(stop), self._st())

    def mlp3_backward_ws_floats(self, net, rows):
        return int(self.lib.smx_mlp3_backward_ws_floats(net.D, net.H1, net.H2, net.OUT, int(rows)))

    def mlp3_dgrad_rows_ws_floats(self, net):
        """floats of the transposed-weights scratch of the fused many-row data gradients (0: shape not supported)"""
        return int(self.lib.smx_mlp3_dgrad_rows_ws_floats(net.D, net.H1, net.H2, net.OUT))

    # ---- fused row-block epoch kernels (csrc/smx_epoch.hip) ------------------------------
    def epoch_supported(self, *nets):
        """every network by itself, and the launch that carries all of them (its LDS tiles are sized for
        the widest layer of any job)"""
        ok = all(self.lib.smx_epoch_supported(n.D, n.H1, n.H2, n.OUT) for n in nets)
        return bool(ok and self.lib.smx_epoch_supported(max(n.D for n in nets), max(n.H1 for n in nets),
                                                        max(n.H2 for n in nets), max(n.OUT for n in nets)))

    def epoch_blocks(self, rows):
        return self.lib.smx_epoch_blocks(rows)

    def epoch_packed_numel(self, net):
        return self.lib.smx_epoch_packed_floats(net.D, net.H1, net.H2, net.OUT)

    def epoch_pack(self, items):
        """items: [(net, packed)], up to 4: the weights in the forward kernel's fragment order, one launch"""
        arr = (L.EpochPack * len(items))()
        for k, (net, packed) in enumerate(items):
            arr[k].net = ctypes.pointer(net.desc)
            arr[k].packed = L.ptr(packed)
        L.call('smx_epoch_pack_f32', arr, len(items), self._st())

    def epoch_prepare(self, obs0, xn, xnT, xr, zmean=None, zstd=None, ref_filter=None, obs_next=None, xnext=None,
                      ref_log_var=None, ref_std=None, pack=(), zero_words=None):
        """the per-learn preparation of the epoch loops in one launch (smx_epoch_prepare_f32): obs0 /
        obs_next are row-strided [rows, D] views, ref_filter the reference policy's ZFilter (or None),
        ref_std the [rows, A] std columns of ref_pol (a column-slice view), pack [(net, packed)]"""
        rows, D = obs0.shape
        a = L.EpochPrep()
        a.obs0, a.ld_obs0, a.rows, a.D = L.ptr(obs0), _row_stride(obs0, D), rows, D
        a.zmean, a.zstd = L.ptr(zmean), L.ptr(zstd)
        a.xn, a.xnT = L.ptr(xn), L.ptr(xnT)
        a.ldT = 0 if xnT is None else (xnT.stride(0) if xnT.shape[0] > 1 else xnT.shape[1])
        if ref_filter is not None:
            a.ref_sum, a.ref_sumsq, a.ref_count = L.ptr(ref_filter.running_sum), L.ptr(ref_filter.running_sumsq), \
                L.ptr(ref_filter.count)
            a.ref_eps, a.ref_filter = float(ref_filter.eps), 1
        a.xr = L.ptr(xr)
        if xnext is not None:
            a.obs_next, a.ld_next, a.xnext = L.ptr(obs_next), _row_stride(obs_next, D), L.ptr(xnext)
        if ref_std is not None:
            A = ref_std.shape[1]
            a.A, a.ref_log_var, a.ref_std, a.ld_ref = A, L.ptr(ref_log_var), L.ptr(ref_std), _row_stride(ref_std, A)
        if zero_words is not None:
            a.zero_words, a.n_zero = L.ptr(zero_words), zero_words.numel()
        a.n_pack = len(pack)
        for k, (net, packed) in enumerate(pack):
            a.pack[k].net = ctypes.pointer(net.desc)
            a.pack[k].packed = L.ptr(packed)
        L.call('smx_epoch_prepare_f32', ctypes.byref(a), self._st())

    _EPOCH_LOSS = {None: L.EPOCH_LOSS_NONE, 'policy': L.EPOCH_LOSS_POLICY, 'value': L.EPOCH_LOSS_VALUE,
                   'rhs_surr': L.EPOCH_RHS_SURR, 'rhs_kl': L.EPOCH_RHS_KL}

    def _epoch_jobs(self, jobs):
        arr = (L.EpochJob * len(jobs))()
        for k, j in enumerate(jobs):
            g = lambda name: L.ptr(j.get(name)) if j.get(name) is not None else None  # noqa: E731
            x = j['x']
            assert x.is_contiguous()
            arr[k].net = ctypes.pointer(j['net'].desc)
            arr[k].x, arr[k].rows = L.ptr(x), x.shape[0]
            arr[k].h1T, arr[k].h2T = g('h1T'), g('h2T')
            tv = j.get('h1T')
            arr[k].ldT = 0 if tv is None else (tv.stride(0) if tv.shape[0] > 1 else tv.shape[1])
            out = j.get('out')
            arr[k].out = g('out')
            arr[k].out_ld = 0 if out is None or out.dim() < 2 or out.stride(0) == out.shape[1] else out.stride(0)
            arr[k].out_act = int(j.get('act', 0))
            arr[k].loss = self._EPOCH_LOSS[j.get('loss')]
            arr[k].stop_flag = g('stop')
            arr[k].dz3, arr[k].dz3T, arr[k].dz2T, arr[k].dz1T = g('dz3'), g('dz3T'), g('dz2T'), g('dz1T')
            arr[k].packed = g('packed')
        return arr

    @staticmethod
    def _epoch_loss(loss):
        if loss is None:
            return None
        a = L.PpoLosses()
        g = lambda name: L.ptr(loss.get(name)) if loss.get(name) is not None else None  # noqa: E731
        a.mode = int(loss.get('mode', 0))
        if loss.get('log_var') is not None:
            A = loss['log_var'].numel()
            a.A, a.rows = A, int(loss['rows'])
            a.log_var, a.adv = g('log_var'), g('adv')
            if loss.get('actions') is not None:
                a.actions, a.ld_act = g('actions'), _row_stride(loss['actions'], A)
                a.behave, a.ld_beh = g('behave'), _row_stride(loss['behave'], 2 * A)
                a.ref, a.ld_ref = g('ref'), _row_stride(loss['ref'], 2 * A)
            a.g_surr, a.g_kl, a.row_partials = g('g_surr'), g('g_kl'), g('partials')
            a.check_stop, a.will_update = int(loss.get('check_stop', 0)), int(loss.get('will_update', 0))
            a.dlogvar, a.dlogvar_sumsq, a.stats = g('dlogvar'), g('dlogvar_sumsq'), g('stats')
        a.returns, a.v_dz3, a.v_partials = g('returns'), g('v_dz3'), g('v_partials')
        a.v_will_update = int(loss.get('v_will_update', 0))
        return ctypes.byref(a)

    def epoch_forward(self, jobs, loss=None, ctrl=None, n_total=0):
        """jobs: dicts(net, x[, h1T, h2T, out, act, loss='policy'|'value', stop]); loss: dict of the
        loss tensors (see smx_epoch_forward_f32).  One launch: 16 rows per workgroup through the three
        layers and the job's loss."""
        L.call('smx_epoch_forward_f32', self._epoch_jobs(jobs), len(jobs), self._epoch_loss(loss),
               L.ptr(ctrl), int(n_total), self._st())

    def epoch_backward(self, jobs, loss, ctrl, n_total):
        """jobs: dicts(net, x, h1T, h2T, loss, dz2T, dz1T[, dz3T (policy), dz3 (value), stop])"""
        L.call('smx_epoch_backward_f32', self._epoch_jobs(jobs), len(jobs), self._epoch_loss(loss),
               L.ptr(ctrl), int(n_total), self._st())

    def epoch_fwdbwd_supported(self, *nets):
        return bool(all(self.lib.smx_epoch_fwdbwd_supported(n.D, n.H1, n.H2, n.OUT) for n in nets) and
                    self.lib.smx_epoch_fwdbwd_supported(max(n.D for n in nets), max(n.H1 for n in nets),
                                                        max(n.H2 for n in nets), max(n.OUT for n in nets)))

    def epoch_fwdbwd(self, jobs, loss, ctrl, n_total, sync_word, kl_slots):
        """epoch_forward + epoch_backward of an updating epoch in ONE launch (smx_epoch_fwdbwd_f32); sync_word: one
        int32, kl_slots: 2 * epoch_blocks(rows) 4-byte words (8-byte aligned) per launch, zero on entry"""
        L.call('smx_epoch_fwdbwd_f32', self._epoch_jobs(jobs), len(jobs), self._epoch_loss(loss),
               L.ptr(ctrl), int(n_total), L.ptr(sync_word), L.ptr(kl_slots), self._st())

    def layernorm_forward(self, x, gamma, beta, eps, y, mean=None, rstd=None):
        """y = LayerNorm(x) over the last dimension (rows of x / y may be strided); mean / rstd [rows] for the backward"""
        rows, F = x.shape
        L.call('smx_layernorm_forward_f32', L.ptr(x), x.stride(0), rows, F, L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(y),
               y.stride(0), L.ptr(mean), L.ptr(rstd), self._st())

    def layernorm_backward_ws_floats(self, rows, F):
        return int(self.lib.smx_layernorm_backward_ws_floats(int(rows), int(F)))

    def layernorm_backward(self, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, ws, relu_mask=False):
        """dx (optionally times x > 0: x is a ReLU's output) and the affine parameters' gradients (overwritten)"""
        rows, F = x.shape
        L.call('smx_layernorm_backward_f32', L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), L.ptr(mean), L.ptr(rstd),
               L.ptr(gamma), rows, F, int(bool(relu_mask)), L.ptr(dx), dx.stride(0), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws),
               ws.numel(), self._st())

    def device_occupy(self, blocks, microseconds):
        """a co-tenant on the current stream: `blocks` workgroups that each hold one CU for `microseconds` (smx_device_occupy)"""
        L.call('smx_device_occupy', int(blocks), int(microseconds), self._st())

    def mlp3_wgrad_multi(self, jobs):
        """the weight-gradient launch alone: dicts(net, x (for rows), grads, sumsq, xT, h1T, h2T, dz3T,
        dz2T, dz1T[, stop])"""
        L.call('smx_mlp3_wgrad_multi_f32', self._jobs(jobs), len(jobs), self._st())

    # ---- GAE / normalisation ------------------------------------------------------------
    def gae(self, values, rewards, dones, gpow, lpow, gamma, gamma_H, B, N, H, adv, ret,
            values_tail=None):
        L.call('smx_windowed_gae_returns_f32', L.ptr(values), L.ptr(values_tail), L.ptr(rewards),
               L.ptr(dones),
               L.ptr(gpow), L.ptr(lpow), float(gamma), float(gamma_H), B, N, H, L.ptr(adv),
               L.ptr(ret), self._st())

    def gae_norm(self, values, rewards, dones, gpow, lpow, gamma, gamma_H, B, N, H, adv, ret, adv_mom, min_std, ticket,
                 values_tail=None):
        """gae + moments + adv_normalize in one launch (single rank)"""
        L.call('smx_windowed_gae_norm_f32', L.ptr(values), L.ptr(values_tail), L.ptr(rewards), L.ptr(dones),
               L.ptr(gpow), L.ptr(lpow), float(gamma), float(gamma_H), B, N, H, L.ptr(adv), L.ptr(ret), L.ptr(adv_mom),
               float(min_std), L.ptr(ticket), self._st())

    def reward_filter_partials(self):
        return int(self.lib.smx_reward_filter_partials())

    def reward_filter(self, rewards, scale, state, eps, out, partials, ticket, use_filter=True, update=True, sums=None):
        """out = clamp((rewards * scale - mean) / std, -5, 5) from `state` = [count, sum, sumsq] as it was BEFORE
        the call, then state <- batch (reward_filter.py:33-57); use_filter=False: the scale alone"""
        L.call('smx_reward_filter_f32', L.ptr(rewards), rewards.numel(), float(scale), int(use_filter), L.ptr(state),
               float(eps), int(update), L.ptr(out), L.ptr(sums), L.ptr(partials), L.ptr(ticket), self._st())

    def learn_epilogue(self, ret, ret_mom, log_var, out4, ticket, zfilter=None, x=None, count_rows=0, v_partials=None,
                       n_epochs=0, nblk=0, v_stats=None, stats_stride=0):
        """value_finalize + moments(ret) + zfilter_update(x) + final_stats in one launch (single rank)"""
        a = L.LearnEpilogue()
        if zfilter is not None:
            rows, D = x.shape
            a.x, a.ldx, a.rows, a.D = L.ptr(x), _row_stride(x, D), rows, D
            a.running_sum, a.running_sumsq, a.count = L.ptr(zfilter.running_sum), L.ptr(zfilter.running_sumsq), \
                L.ptr(zfilter.count)
            a.count_rows = float(count_rows)
        a.A = log_var.numel()
        a.ret, a.n_ret, a.ret_moments = L.ptr(ret), ret.numel(), L.ptr(ret_mom)
        a.v_partials, a.n_epochs, a.nblk, a.v_stats, a.stats_stride = L.ptr(v_partials), n_epochs, nblk, L.ptr(v_stats), \
            stats_stride
        a.log_var, a.out4, a.ticket = L.ptr(log_var), L.ptr(out4), L.ptr(ticket)
        L.call('smx_ppo_learn_epilogue_f32', ctypes.byref(a), self._st())

    def moments(self, x, out):
        L.call('smx_moments_f32', L.ptr(x), x.numel(), L.ptr(out), self._st())

    def moments_merge(self, parts, out):
        L.call('smx_moments_merge_f32', L.ptr(parts), parts.numel() // 3, L.ptr(out), self._st())

    def adv_normalize(self, x, mom, min_std):
        L.call('smx_adv_normalize_f32', L.ptr(x), x.numel(), L.ptr(mom), float(min_std), self._st())

    # ---- losses -------------------------------------------------------------------------
    def loss_blocks(self, rows):
        return self.lib.smx_ppo_loss_blocks(rows)

    def policy_loss(self, mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl,
                    partials):
        rows, A = mean.shape
        L.call('smx_ppo_policy_loss_f32', mode, L.ptr(mean), L.ptr(log_var), L.ptr(actions),
               _row_stride(actions, A), L.ptr(behave), _row_stride(behave, 2 * A), L.ptr(ref),
               _row_stride(ref, 2 * A), L.ptr(adv), rows, A, L.ptr(ctrl), L.ptr(g_surr),
               L.ptr(g_kl), L.ptr(partials), self._st())

    def partials_fold(self, partials, nblk, out, ctrl=None):
        """out[j] = sum of the partial rows [j R, (j + 1) R), R = ceil(nblk / len(out)) (smx_ppo_partials_fold_f32)"""
        L.call('smx_ppo_partials_fold_f32', L.ptr(partials), int(nblk), partials.stride(0), L.ptr(out), out.shape[0],
               L.ptr(ctrl), self._st())

    def policy_finalize(self, mode, partials, nblk, g_surr, g_kl, log_var, n_total, ctrl,
                        check_stop, will_update, dz3, dlogvar, dlogvar_sumsq, stats, dz3_t=None):
        rows, A = g_surr.shape
        L.call('smx_ppo_loss_finalize_f32', mode, L.ptr(partials), nblk, L.ptr(g_surr),
               L.ptr(g_kl), L.ptr(log_var), rows, n_total, A, L.ptr(ctrl), int(check_stop),
               int(will_update), L.ptr(dz3), L.ptr(dz3_t),
               0 if dz3_t is None else dz3_t.stride(0), L.ptr(dlogvar), L.ptr(dlogvar_sumsq),
               L.ptr(stats), self._st())

    def _losses_args(self, mode, mean, log_var, actions, behave, ref, adv, g_surr, g_kl, partials,
                     check_stop, will_update, values, returns, v_dz3, v_partials, v_will_update):
        rows, A = mean.shape
        a = L.PpoLosses()
        a.mode, a.A, a.rows = mode, A, rows
        a.mean, a.log_var, a.adv = L.ptr(mean), L.ptr(log_var), L.ptr(adv)
        a.actions, a.ld_act = L.ptr(actions), _row_stride(actions, A)
        a.behave, a.ld_beh = L.ptr(behave), _row_stride(behave, 2 * A)
        a.ref, a.ld_ref = L.ptr(ref), _row_stride(ref, 2 * A)
        a.g_surr, a.g_kl, a.row_partials = L.ptr(g_surr), L.ptr(g_kl), L.ptr(partials)
        a.check_stop, a.will_update = int(check_stop), int(will_update)
        a.values, a.returns = L.ptr(values), L.ptr(returns)
        a.v_dz3, a.v_partials, a.v_will_update = L.ptr(v_dz3), L.ptr(v_partials), int(v_will_update)
        return a

    def epoch_losses(self, mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl,
                     partials, check_stop, will_update, dz3, dlogvar, dlogvar_sumsq, stats, dz3_t=None,
                     values=None, returns=None, v_dz3=None, v_partials=None, v_will_update=True):
        """policy_loss + policy_finalize (+ value_loss) of a single-GPU lock-step epoch, one launch"""
        a = self._losses_args(mode, mean, log_var, actions, behave, ref, adv, g_surr, g_kl, partials,
                              check_stop, will_update, values, returns, v_dz3, v_partials, v_will_update)
        a.dz3, a.dz3_t = L.ptr(dz3), L.ptr(dz3_t)
        a.ld_t = 0 if dz3_t is None else (dz3_t.stride(0) if dz3_t.shape[0] > 1 else dz3_t.shape[1])
        a.dlogvar, a.dlogvar_sumsq, a.stats = L.ptr(dlogvar), L.ptr(dlogvar_sumsq), L.ptr(stats)
        L.call('smx_ppo_epoch_losses_f32', ctypes.byref(a), L.ptr(ctrl), self._st())

    def epoch_losses_dp(self, mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl,
                        partials, n_total, g_surr_t=None, g_kl_t=None, values=None, returns=None,
                        v_dz3=None, v_partials=None, v_will_update=True):
        """the loss launch of a data-parallel lock-step epoch: gradient tiles come out divided by
        n_total (+ transposed copies), no finalize -- see smx_ppo_epoch_combine_f32"""
        a = self._losses_args(mode, mean, log_var, actions, behave, ref, adv, g_surr, g_kl, partials,
                              False, False, values, returns, v_dz3, v_partials, v_will_update)
        t = g_surr_t
        a.ld_t = 0 if t is None else (t.stride(0) if t.shape[0] > 1 else t.shape[1])
        L.call('smx_ppo_epoch_losses_dp_f32', ctypes.byref(a), int(n_total), L.ptr(g_surr_t),
               L.ptr(g_kl_t), L.ptr(ctrl), self._st())

    def epoch_combine(self, mode, partials, nblk, n_total, log_var, ctrl, check_stop, will_update,
                      stats, grads_a, grads_kl, n_mlp, sumsq_a, grads_c=None, sumsq_c=None):
        """after the all-reduce of a data-parallel epoch: grads_a += c_kl * grads_kl, log_var's
        gradient, statistics / early exit, sum-of-squares partials of both groups"""
        a = L.PpoCombine()
        a.mode, a.A, a.nblk, a.n_total = mode, log_var.numel(), nblk, int(n_total)
        a.row_partials, a.log_var, a.stats = L.ptr(partials), L.ptr(log_var), L.ptr(stats)
        a.check_stop, a.will_update = int(check_stop), int(will_update)
        a.grads_a, a.grads_kl = L.ptr(grads_a), L.ptr(grads_kl)
        a.n_mlp, a.n_a, a.sumsq_a = int(n_mlp), grads_a.numel(), L.ptr(sumsq_a)
        a.grads_c, a.sumsq_c = L.ptr(grads_c), L.ptr(sumsq_c)
        a.n_c = 0 if grads_c is None else grads_c.numel()
        L.call('smx_ppo_epoch_combine_f32', ctypes.byref(a), L.ptr(ctrl), self._st())

    def final_stats(self, log_var, zfilter, out4):
        """means the learner reports once per learn (ppo.py:572, 580-583), formed on the device so
        that the statistics need ONE read-back"""
        zf = zfilter
        L.call('smx_ppo_final_stats_f32', L.ptr(log_var), log_var.numel(),
               L.ptr(zf.running_sum) if zf is not None else None,
               L.ptr(zf.running_sumsq) if zf is not None else None,
               L.ptr(zf.count) if zf is not None else None,
               zf.running_sum.numel() if zf is not None else 0, L.ptr(out4), self._st())

    def value_loss_blocks(self, rows):
        return self.lib.smx_value_loss_blocks(rows)

    def value_loss(self, values, returns, n_total, dz3, partials, ctrl, will_update):
        L.call('smx_value_loss_f32', L.ptr(values), L.ptr(returns), values.numel(), n_total,
               L.ptr(dz3), L.ptr(partials), L.ptr(ctrl), int(will_update), self._st())

    def value_finalize(self, partials, count, nblk, stats, stride):
        L.call('smx_value_loss_finalize_f32', L.ptr(partials), count, nblk, L.ptr(stats), stride,
               self._st())

    # ---- optimiser ----------------------------------------------------------------------
    def clip_adam(self, theta, grads, m, v, sumsq, npart, ctrl, which, honour_stop, grad_norm_out, pack=None):
        if pack is not None:
            g = L.AdamGroup(L.ptr(theta), L.ptr(grads), L.ptr(m), L.ptr(v), theta.numel(), L.ptr(sumsq), npart,
                            int(honour_stop), L.ptr(grad_norm_out))
            g.pack_net, g.packed = ctypes.pointer(pack[0].desc), L.ptr(pack[1])
            L.call('smx_clip_adam_step_group_f32', ctypes.byref(g), which, L.ptr(ctrl), self._st())
            return
        L.call('smx_clip_adam_step_f32', L.ptr(theta), L.ptr(grads), L.ptr(m), L.ptr(v),
               theta.numel(), L.ptr(sumsq), npart, L.ptr(ctrl), which, int(honour_stop),
               L.ptr(grad_norm_out), self._st())

    def clip_adam_pair(self, actor, critic, ctrl, pack=None):
        """actor / critic: (theta, grads, m, v, sumsq, npart, honour_stop, grad_norm_out);
        pack: ((actor net, packed), (critic net, packed)) -- the step also refreshes the fused epoch
        kernels' packed weight copies"""
        gs = []
        for k, (theta, grads, m, v, sumsq, npart, honour_stop, gno) in enumerate((actor, critic)):
            g = L.AdamGroup(L.ptr(theta), L.ptr(grads), L.ptr(m), L.ptr(v), theta.numel(),
                            L.ptr(sumsq), npart, int(honour_stop), L.ptr(gno))
            if pack is not None:
                g.pack_net, g.packed = ctypes.pointer(pack[k][0].desc), L.ptr(pack[k][1])
            gs.append(g)
        L.call('smx_clip_adam_step_pair_f32', ctypes.byref(gs[0]), ctypes.byref(gs[1]), L.ptr(ctrl),
               self._st())

    def sumsq_blocks(self, n):
        return self.lib.smx_sumsq_blocks(n)

    def sumsq_partials(self, x, partials):
        L.call('smx_sumsq_partials_f32', L.ptr(x), x.numel(), L.ptr(partials), self._st())

    # ---- replay / windowing ---------------------------------------------------------------
    def ring_insert(self, table, cursor, src):
        """table [capacity, width] <- src [n, width] at ring position `cursor`; fp32 or any other element type
        (uint8 frames), as long as both agree"""
        cap, width = table.shape
        assert table.dtype == src.dtype and src.is_contiguous(), (table.dtype, src.dtype)
        if table.dtype == torch.float32:
            L.call('smx_ring_insert_f32', L.ptr(table), cap, width, int(cursor), L.ptr(src),
                   src.shape[0], self._st())
        else:
            L.call('smx_ring_insert_bytes', L.ptr(table), cap, width * table.element_size(), int(cursor),
                   L.ptr(src), src.shape[0], self._st())

    def gather_rows(self, table, idx, dst):
        cap, width = table.shape
        assert table.dtype == dst.dtype and dst.is_contiguous(), (table.dtype, dst.dtype)
        if table.dtype == torch.float32:
            L.call('smx_gather_rows_f32', L.ptr(table), cap, width, L.ptr(idx), idx.numel(),
                   L.ptr(dst), self._st())
        else:
            L.call('smx_gather_rows_bytes', L.ptr(table), cap, width * table.element_size(), L.ptr(idx),
                   idx.numel(), L.ptr(dst), self._st())

    def uniform_gather_multi(self, tables, outs, length, seed, offset, idx=None, idx_out=None):
        """a whole uniform sample in one launch: row idx[i] -- or, idx None, the row smx_uniform_indices(length, seed,
        offset) would draw -- of every table [capacity, width] -> outs[k] [rows, width] (<= 8 tables, fp32 or uint8)"""
        arr = (L.GatherJob * len(tables))()
        cap, rows = tables[0].shape[0], outs[0].shape[0]
        for k, (t, o) in enumerate(zip(tables, outs)):
            assert t.dtype == o.dtype and o.is_contiguous() and t.shape[0] == cap and o.shape[0] == rows
            arr[k].table, arr[k].dst, arr[k].row_bytes = L.ptr(t), L.ptr(o), t.shape[1] * t.element_size()
        L.call('smx_uniform_gather_multi', arr, len(tables), cap, rows, L.ptr(idx), int(length), int(seed), int(offset),
               L.ptr(idx_out), self._st())

    def philox4x32_10(self, ctr_key, out):
        """ctr_key [n, 6] int32 (bit patterns of uint32) -> out [n, 4] int32: the sampler's generator by itself"""
        L.call('smx_philox4x32_10', L.ptr(ctr_key), ctr_key.shape[0], L.ptr(out), self._st())

    def uniform_indices(self, idx, length, seed, offset):
        L.call('smx_uniform_indices', L.ptr(idx), idx.numel(), int(length), int(seed), int(offset),
               self._st())

    def window_emit(self, src, start, n_step, stride, W, dst):
        """src [actors, T, width] -> dst [actors*W, n_step, width]"""
        actors, T, width = src.shape
        assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype
        if src.dtype == torch.float32:
            L.call('smx_window_emit_f32', L.ptr(src), actors, T, width, start, n_step, stride, W,
                   L.ptr(dst), self._st())
        else:
            L.call('smx_window_emit_bytes', L.ptr(src), actors, T, width * src.element_size(), start, n_step,
                   stride, W, L.ptr(dst), self._st())

    def frame_stack(self, frames, n_stack, start, n_step, stride, W, dst, episode_first=None):
        """frames [actors, R, C, H, W] uint8 (one raw frame per step) -> dst [actors*W, n_step, n_stack*C, H, W]: the
        last n_stack frames on the channel axis, oldest first, history filled with the first frame after a reset"""
        actors, R = frames.shape[:2]
        fb = frames[0, 0].numel() * frames.element_size()
        assert frames.is_contiguous() and dst.is_contiguous() and frames.dtype == dst.dtype
        assert dst.numel() == actors * W * n_step * n_stack * frames[0, 0].numel()
        L.call('smx_frame_stack_u8', L.ptr(frames), actors, R, fb, int(n_stack), L.ptr(episode_first), int(start),
               int(n_step), int(stride), int(W), L.ptr(dst), self._st())

    def synth_frames(self, s0, t, dst):
        """the synthetic camera for all actors: s0 [n] (strided view allowed), dst [n, C, H, W] uint8 (row-strided
        view allowed: a rollout slot)"""
        n, C, H, W = dst.shape
        assert dst.dtype == torch.uint8 and dst[0].is_contiguous() and s0.dim() == 1
        L.call('smx_synth_frame_u8', L.ptr(s0), s0.stride(0), n, C, H, W, int(t), L.ptr(dst), dst.stride(0), self._st())

    @staticmethod
    def _synth_act_step(state, init_state, mean, A, log_var, noise_scale, eps, t, episode_len, slot, rolls, zfilter,
                        xn_out):
        n, D = state.shape
        p = L.SynthActStep()
        p.state, p.init_state = L.ptr(state), L.ptr(init_state)
        p.mean, p.ld_mean, p.log_var = L.ptr(mean), (0 if mean is None else _row_stride(mean, A)), L.ptr(log_var)
        p.noise_scale, p.eps = L.ptr(noise_scale), L.ptr(eps)
        p.ld_eps = 0 if eps is None else _row_stride(eps, A)
        p.n, p.D, p.A, p.t, p.episode_len, p.slot = n, D, A, int(t), int(episode_len), int(slot)
        r = rolls or {}
        p.T = r['obs'].shape[1] if 'obs' in r else 1
        p.obs_roll, p.act_roll = L.ptr(r.get('obs')), L.ptr(r.get('actions'))
        p.rew_roll, p.done_roll, p.pd_roll = L.ptr(r.get('rewards')), L.ptr(r.get('dones')), L.ptr(r.get('pds'))
        if zfilter is not None:
            p.zsum, p.zsumsq, p.zcount = L.ptr(zfilter.running_sum), L.ptr(zfilter.running_sumsq), L.ptr(zfilter.count)
            p.zeps = float(zfilter.eps)
        p.xn_out = L.ptr(xn_out)
        return p

    def synth_act_env_step(self, state, init_state, mean, log_var, noise_scale, eps, t, episode_len,
                           slot, rolls, zfilter, xn_out):
        """acting head + env step + next observation's z-filter, one launch (see the header);
        rolls: dict obs / actions / rewards / dones [/ pds] or None; zfilter: ZFilter or None"""
        p = self._synth_act_step(state, init_state, mean, mean.shape[1], log_var, noise_scale, eps, t, episode_len,
                                 slot, rolls, zfilter, xn_out)
        L.call('smx_synth_act_env_step_f32', ctypes.byref(p), self._st())

    def synth_act_env_step_head(self, W3, b3, h2, out_act, state, init_state, log_var, noise_scale, eps, t,
                                episode_len, slot, rolls, zfilter, xn_out):
        """the same launch with the policy's output layer folded in: mean = act(h2 . W3^T + b3) formed per actor"""
        A, H2 = W3.shape
        p = self._synth_act_step(state, init_state, None, A, log_var, noise_scale, eps, t, episode_len, slot, rolls,
                                 zfilter, xn_out)
        L.call('smx_synth_act_env_step_head_f32', ctypes.byref(p), L.ptr(W3), L.ptr(b3), L.ptr(h2),
               _row_stride(h2, H2), H2, int(out_act), self._st())

    def synth_rollout_supported(self, net):
        return bool(self.lib.smx_synth_rollout_supported(net.D, net.H1, net.H2, net.OUT))

    def synth_rollout(self, net, packed, out_act, state, init_state, log_var, noise_scale, eps, t, episode_len,
                      steps, slot, rolls, zfilter):
        """`steps` acting + environment steps of all actors in ONE launch (csrc/smx_rollout.hip): a workgroup owns
        16 actors for the whole rollout.  packed: epoch_pack of `net`; eps [steps, n, A] or None; rolls as in
        synth_act_env_step ([n, T + 1, .] tables)."""
        n, D = state.shape
        p = L.SynthRollout()
        p.net, p.packed, p.out_act, p.n = ctypes.pointer(net.desc), L.ptr(packed), int(out_act), n
        p.log_var, p.noise_scale, p.eps = L.ptr(log_var), L.ptr(noise_scale), L.ptr(eps)
        if eps is not None:
            assert eps.is_contiguous() and tuple(eps.shape) == (steps, n, net.OUT)
        if zfilter is not None:
            p.zsum, p.zsumsq, p.zcount = L.ptr(zfilter.running_sum), L.ptr(zfilter.running_sumsq), L.ptr(zfilter.count)
            p.zeps = float(zfilter.eps)
        r = rolls or {}
        p.t, p.episode_len, p.steps, p.slot = int(t), int(episode_len), int(steps), int(slot)
        p.rows_per_actor = r['obs'].shape[1] if 'obs' in r else 1
        p.state, p.init_state = L.ptr(state), L.ptr(init_state)
        p.obs_roll, p.act_roll = L.ptr(r.get('obs')), L.ptr(r.get('actions'))
        p.rew_roll, p.done_roll, p.pd_roll = L.ptr(r.get('rewards')), L.ptr(r.get('dones')), L.ptr(r.get('pds'))
        p.obs_last = L.ptr(r.get('obs_last'))          # rows_per_actor == steps: the replay's layout (obs_next apart)
        L.call('smx_synth_rollout_f32', ctypes.byref(p), self._st())

    def synth_env_step(self, state, init_state, actions, t, episode_len, slot, obs_roll, act_roll,
                       rew_roll, done_roll):
        n, D = state.shape
        A = actions.shape[1]
        T = obs_roll.shape[1] if obs_roll is not None else 1
        L.call('smx_synth_env_step_f32', L.ptr(state), L.ptr(init_state), L.ptr(actions), n, D, A,
               int(t), int(episode_len), int(slot), T, L.ptr(obs_roll), L.ptr(act_roll),
               L.ptr(rew_roll), L.ptr(done_roll), self._st())


    # ---- generic dense layer + DDPG pieces ---------------------------------------------------
    def linear(self, A, a_kc, B, b_kc, bias, C, M, N, K, act=0, relu_mask=None, lda=None, ldb=None,
               ldc=None, stop=None):
        """C[M,N] = act(A . B^T + bias) with explicit leading strides (views into wider buffers)"""
        lda = lda if lda is not None else A.stride(0)
        ldb = ldb if ldb is not None else B.stride(0)
        ldc = ldc if ldc is not None else C.stride(0)
        L.call('smx_linear_f32', L.ptr(A), lda, int(a_kc), L.ptr(B), ldb, int(b_kc), L.ptr(bias),
               L.ptr(C), ldc, M, N, K, act, L.ptr(relu_mask), L.ptr(stop), self._st())

    def linear_multi(self, jobs):
        """independent dense problems in ONE launch (<= 9).  Each job is the argument tuple of ``linear``
        (``('linear', A, a_kc, B, b_kc, bias, C, M, N, K, {act, relu_mask, lda, ldb, ldc, stop})``) or of
        ``linear_wgrad`` (``('wgrad', dZ, X, dW, db, M, N, rows, {ldz, ldx, ldw})``)."""
        arr = (L.LinearJob * len(jobs))()
        for k, j in enumerate(jobs):
            kw = j[-1] if isinstance(j[-1], dict) else {}
            a = arr[k]
            if j[0] == 'wgrad':
                _, dZ, X, dW, db, M, N, rows = j[:8]
                a.kind, a.A, a.B, a.C, a.dbias = 1, L.ptr(dZ), L.ptr(X), L.ptr(dW), L.ptr(db)
                a.lda, a.ldb, a.ldc = kw.get('ldz') or dZ.stride(0), kw.get('ldx') or X.stride(0), kw.get('ldw') or dW.stride(0)
                a.M, a.N, a.K = M, N, rows
            else:
                _, A, a_kc, B, b_kc, bias, C, M, N, K = j[:10]
                a.kind, a.act = 0, int(kw.get('act', 0))
                a.A, a.B, a.bias, a.C, a.relu_mask = L.ptr(A), L.ptr(B), L.ptr(bias), L.ptr(C), L.ptr(kw.get('relu_mask'))
                a.lda, a.ldb, a.ldc = kw.get('lda') or A.stride(0), kw.get('ldb') or B.stride(0), kw.get('ldc') or C.stride(0)
                a.a_kcontig, a.b_kcontig, a.M, a.N, a.K = int(a_kc), int(b_kc), M, N, K
                a.stop_flag = L.ptr(kw.get('stop'))
        L.call('smx_linear_multi_f32', arr, len(jobs), self._st())

    def linear_wgrad(self, dZ, X, dW, db, M, N, rows, ldz=None, ldx=None, ldw=None, ws=None):
        """ws: optional split-K workspace (>= linear_wgrad_ws_floats(M, N, rows) floats)"""
        ldz = ldz if ldz is not None else dZ.stride(0)
        ldx = ldx if ldx is not None else X.stride(0)
        ldw = ldw if ldw is not None else dW.stride(0)
        L.call('smx_linear_wgrad_splitk_f32', L.ptr(dZ), ldz, L.ptr(X), ldx, L.ptr(dW), ldw, L.ptr(db),
               M, N, rows, L.ptr(ws), 0 if ws is None else ws.numel(), self._st())

    def linear_wgrad_pair(self, dZ, X1, dW1, db1, X2, dW2, db2, rows, ws):
        """dW1 = dZ^T . X1, dW2 = dZ^T . X2 (+ column sums of dZ) -- one split-K launch where both run on the 32 x 32 kernel;
        ws >= linear_wgrad_ws_floats(M, N1, rows) + linear_wgrad_ws_floats(M, N2, rows) floats"""
        M, N1, N2 = dW1.shape[0], dW1.shape[1], dW2.shape[1]
        L.call('smx_linear_wgrad_splitk_pair_f32', L.ptr(dZ), dZ.stride(0), M, rows, L.ptr(X1), X1.stride(0), L.ptr(dW1),
               L.ptr(db1), N1, L.ptr(X2), X2.stride(0), L.ptr(dW2), L.ptr(db2), N2, L.ptr(ws),
               0 if ws is None else ws.numel(), self._st())

    def linear_wgrad_ws_floats(self, M, N, rows):
        return int(self.lib.smx_linear_wgrad_ws_floats(M, N, rows))

    def lstm_backward_ws_floats(self, net, B, T):
        return int(self.lib.smx_lstm_backward_ws_floats(net.D, net.H, B, T))

    def ddpg_critic_loss(self, q, q_next, rewards, dones, gamma_n, y, dz3):
        L.call('smx_ddpg_critic_loss_f32', L.ptr(q), L.ptr(q_next), L.ptr(rewards), L.ptr(dones),
               float(gamma_n), q.numel(), L.ptr(y), L.ptr(dz3), self._st())

    def tanh_backward(self, da, a, out):
        L.call('smx_tanh_backward_f32', L.ptr(da), L.ptr(a), a.numel(), L.ptr(out), self._st())

    def fill(self, x, value):
        L.call('smx_fill_f32', L.ptr(x), x.numel(), float(value), self._st())

    def adam_step(self, theta, grads, m, v, lr, step, weight_decay=0.0, clip_value=0.0):
        L.call('smx_adam_step_f32', L.ptr(theta), L.ptr(grads), L.ptr(m), L.ptr(v), theta.numel(),
               float(lr), int(step), float(weight_decay), float(clip_value), self._st())

    def soft_update(self, target, source, tau):
        L.call('smx_soft_update_f32', L.ptr(target), L.ptr(source), float(tau), target.numel(),
               self._st())

    def ddpg_critic_loss_step(self, q, q_next, rewards, dones, gamma_n, y, dz3, step):
        """ddpg_critic_loss + (step[0] += 1): the iteration's Adam step count, on the device"""
        L.call('smx_ddpg_critic_loss_step_f32', L.ptr(q), L.ptr(q_next), L.ptr(rewards), L.ptr(dones),
               float(gamma_n), q.numel(), L.ptr(y), L.ptr(dz3), L.ptr(step), self._st())

    def adam_step_dev(self, theta, grads, m, v, lr, step, weight_decay=0.0, clip_value=0.0):
        """adam_step with lr ([1] float tensor) and step ([1] int32 tensor) read on the device"""
        L.call('smx_adam_step_dev_f32', L.ptr(theta), L.ptr(grads), L.ptr(m), L.ptr(v), theta.numel(),
               L.ptr(lr), L.ptr(step), float(weight_decay), float(clip_value), self._st())

    def hard_update_every(self, target, source, step, interval):
        L.call('smx_hard_update_every_f32', L.ptr(target), L.ptr(source), target.numel(), L.ptr(step),
               int(interval), self._st())

    def ddpg_stats(self, q, y, rewards, actions, q_actor, stats):
        rows, A = actions.shape
        L.call('smx_ddpg_stats_f32', L.ptr(q), L.ptr(y), L.ptr(rewards), L.ptr(actions),
               _row_stride(actions, A), A, L.ptr(q_actor), rows, L.ptr(stats), self._st())


    # ---- one DDPG iteration on row blocks (smx_ddpg_rows.hip) -------------------------------------------
    def ddpg_rows_supported(self, D, A, H1, H2, c1, c2, rows=None):
        if rows is not None:
            return bool(self.lib.smx_ddpg_rows_supported_at(D, A, H1, H2, c1, c2, int(rows)))
        return bool(self.lib.smx_ddpg_rows_supported(D, A, H1, H2, c1, c2))

    def ddpg_rows_packed_floats(self, D, A, H1, H2, c1, c2):
        return int(self.lib.smx_ddpg_rows_packed_floats(D, A, H1, H2, c1, c2))

    def ddpg_rows_args(self, dims, nets, packed, io, gamma_n):
        """the argument block of the three launches below.  dims = (D, A, H1, H2, c1, c2); nets: {'actor' | 'critic' |
        'target_actor' | 'target_critic': {'W1', 'b1', ... 'b3'}} (views into the parameter buffers); packed: the
        fragment-order weight copy (ddpg_rows_packed_floats floats); io: the tensors smx_ddpg_rows_t names
        (include/surreal_amd.h).  Holds references to every tensor."""
        a = L.DdpgRows()
        a.D, a.A, a.H1, a.H2, a.c1, a.c2 = [int(v) for v in dims]
        a.rows = int(io['x'].shape[0])
        for name in ('actor', 'critic', 'target_actor', 'target_critic'):
            n = getattr(a, name)
            for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3'):
                setattr(n, k, nets[name][k].data_ptr())
        a.packed = packed.data_ptr()
        a.gamma_n = float(gamma_n)
        for k in ('x', 'x_next', 'actions', 'rewards', 'dones', 'xcat', 'h2c', 'q', 'q_next', 'y', 'dz3', 'dz2', 'dxcat',
                  'h1a', 'h2a', 'act', 'q_actor', 'dz3a', 'dz2a', 'dz1a', 'step'):
            t = io.get(k)
            assert t is None or t.is_contiguous(), k
            setattr(a, k, None if t is None else t.data_ptr())
        a._refs = (nets, packed, io)
        return a

    def ddpg_rows_pack(self, args, critic_only=False):
        L.call('smx_ddpg_rows_pack_f32', ctypes.byref(args), 1 if critic_only else 0, self._st())

    def ddpg_rows_critic(self, args):
        L.call('smx_ddpg_rows_critic_f32', ctypes.byref(args), self._st())

    def ddpg_rows_actor(self, args):
        L.call('smx_ddpg_rows_actor_f32', ctypes.byref(args), self._st())

    def ddpg_rows_update(self, args, group, theta, grads, exp_avg, exp_avg_sq, lr, step, weight_decay, clip_value,
                         target=None, tau=0.0, interval=0, wgrad=False, stats=None, stats_host=None):
        """smx_ddpg_rows_update_f32 (wgrad: smx_ddpg_rows_wgrad_update_f32, which forms the gradients first): Adam on the group ('actor' | 'critic'), its target network's update (soft with tau,
        or hard every `interval` iterations of *step; target None: none) and the fragment-order copies of both"""
        u = L.DdpgUpdate()
        u.theta, u.grads, u.exp_avg, u.exp_avg_sq = (t.data_ptr() for t in (theta, grads, exp_avg, exp_avg_sq))
        u.target = None if target is None else target.data_ptr()
        u.n = theta.numel()
        assert grads.numel() == u.n and exp_avg.numel() == u.n and exp_avg_sq.numel() == u.n
        assert target is None or target.numel() == u.n
        u.lr, u.step = lr.data_ptr(), step.data_ptr()
        assert stats is None or wgrad
        u.stats = None if stats is None else stats.data_ptr()
        assert stats_host is None or (stats is not None and stats_host.is_pinned() and stats_host.numel() >= 16)
        u.stats_host = None if stats_host is None else stats_host.data_ptr()
        u.weight_decay, u.clip_value, u.tau, u.interval = float(weight_decay or 0.0), float(clip_value or 0.0), float(tau), int(interval)
        L.call('smx_ddpg_rows_wgrad_update_f32' if wgrad else 'smx_ddpg_rows_update_f32', ctypes.byref(args),
               {'actor': 0, 'critic': 1}[group], ctypes.byref(u), self._st())


    # ---- LSTM stem ----------------------------------------------------------------------------
    def lstm_forward(self, net, x, B, T, h0, c0, gates, out, cs, hprev=None, hN=None, cN=None,
                     stop=None):
        """net: model.ppo_net.LstmParams; x [B*T, D] contiguous; see smx_lstm_forward_f32"""
        L.call('smx_lstm_forward_f32', ctypes.byref(net.desc), L.ptr(x), B, T, L.ptr(h0), L.ptr(c0),
               L.ptr(gates), L.ptr(out), L.ptr(cs), L.ptr(hprev), L.ptr(hN), L.ptr(cN), L.ptr(stop),
               self._st())

    def lstm_backward(self, net, x, B, T, c0, gates, cs, hprev, dout, dgates, grads, stop=None,
                      ws=None):
        L.call('smx_lstm_backward_f32', ctypes.byref(net.desc), L.ptr(x), B, T, L.ptr(c0),
               L.ptr(gates), L.ptr(cs), L.ptr(hprev), L.ptr(dout), L.ptr(dgates), L.ptr(grads),
               L.ptr(stop), L.ptr(ws), 0 if ws is None else ws.numel(), self._st())


    # ---- CNN stem data movement ----------------------------------------------------------------
    def im2col(self, src, F, C, Hin, Win, k, stride, cols, channel_last=False, scale_div=0.0):
        """src: uint8 / fp32 frames [F, C, Hin, Win] or (channel_last) fp32 [F, Hin*Win, C]"""
        L.call('smx_im2col_f32', L.ptr(src), int(src.dtype == torch.uint8), int(channel_last), F, C,
               Hin, Win, k, k, stride, float(scale_div), L.ptr(cols), self._st())

    @staticmethod
    def conv_u8_supported(frames, C, Hin, Win, k, stride, cout, W=None):
        """shapes smx_conv_u8_forward_f32 takes (the implicit-GEMM first convolution over uint8 frames); `W`:
        the layer's weight, a view into the flat parameter buffer whose 16-byte alignment depends on its offset"""
        K = C * k * k
        return frames.dtype == torch.uint8 and cout <= 16 and k % 4 == 0 and Win % 4 == 0 and stride % 4 == 0 \
            and K % 64 == 0 and K <= 256 and frames.data_ptr() % 4 == 0 and (W is None or W.data_ptr() % 16 == 0)

    def conv_u8_forward(self, frames, F, C, Hin, Win, k, stride, W, bias, cout, y, stop=None):
        """y [F*Ho*Wo, cout] = relu(conv(frames / 255, W) + bias), no patch matrix (see the header)"""
        L.call('smx_conv_u8_forward_f32', L.ptr(frames), F, C, Hin, Win, k, stride, L.ptr(W), L.ptr(bias), cout,
               L.ptr(y), L.ptr(stop), self._st())

    def conv_u8_wgrad_ws_floats(self, cout, K):
        return int(self.lib.smx_conv_u8_wgrad_ws_floats(cout, K))

    def conv_u8_wgrad(self, frames, F, C, Hin, Win, k, stride, dy, cout, dW, db, ws, stop=None):
        """dW [cout, C*k*k] (torch's Conv2d.weight order), db [cout] from dy [F*Ho*Wo, cout] and the uint8 frames"""
        L.call('smx_conv_u8_wgrad_f32', L.ptr(frames), F, C, Hin, Win, k, stride, L.ptr(dy), cout, L.ptr(dW), L.ptr(db),
               L.ptr(ws), ws.numel(), L.ptr(stop), self._st())

    @staticmethod
    def conv_cl_supported(src, C, k, cout):
        """shapes smx_conv_cl_forward_f32 / _wgrad_f32 take (implicit GEMMs over a 16-channel channel-last source)"""
        return src.dtype == torch.float32 and C == 16 and cout <= 32 and k in (2, 3, 4) and src.data_ptr() % 16 == 0

    def conv_cl_forward(self, src, F, C, Hin, Win, k, stride, W, bias, cout, y, stop=None):
        L.call('smx_conv_cl_forward_f32', L.ptr(src), F, C, Hin, Win, k, stride, L.ptr(W), L.ptr(bias), cout, L.ptr(y),
               L.ptr(stop), self._st())

    def conv_cl_wgrad_ws_floats(self, cout, k):
        return int(self.lib.smx_conv_cl_wgrad_ws_floats(cout, k))

    def conv_cl_wgrad(self, src, F, C, Hin, Win, k, stride, dy, cout, dW, db, ws, stop=None):
        L.call('smx_conv_cl_wgrad_f32', L.ptr(src), F, C, Hin, Win, k, stride, L.ptr(dy), cout, L.ptr(dW), L.ptr(db),
               L.ptr(ws), ws.numel(), L.ptr(stop), self._st())

    @staticmethod
    def conv_cl_dgrad_supported(dy, C, k, stride, cout):
        return C == 16 and cout <= 32 and k == 2 * stride and dy.data_ptr() % 16 == 0

    def conv_cl_dgrad(self, dy, F, C, Hin, Win, k, stride, W, cout, relu_of, dx, stop=None):
        """dx [F*Hin*Win, 16] = relu'(relu_of) * conv_transpose(dy [F*Ho*Wo, cout], W [cout, 16, k, k])"""
        L.call('smx_conv_cl_dgrad_f32', L.ptr(dy), F, C, Hin, Win, k, stride, L.ptr(W), cout, L.ptr(relu_of), L.ptr(dx),
               L.ptr(stop), self._st())

    def col2im(self, dcols, F, C, Hin, Win, k, stride, relu_of, dx):
        L.call('smx_col2im_f32', L.ptr(dcols), F, C, Hin, Win, k, k, stride, L.ptr(relu_of),
               L.ptr(dx), self._st())

    def flatten_order(self, src, O, C, P, to_channel_last, out):
        L.call('smx_flatten_order_f32', L.ptr(src), O, C, P, int(to_channel_last), L.ptr(out),
               self._st())


# ------------------------------------------------------------------------------------------
# process-wide default.  The product default is HipKernels on 'cuda' and nothing in
# surreal_amd/ ever installs anything else; tests/ install a CPU test double to exercise
# the host logic (sharding, epoch control, stats) without a GPU.
# ------------------------------------------------------------------------------------------
_default = {'kernels': None, 'device': 'cuda'}


def default_kernels():
    if _default['kernels'] is None:
        _default['kernels'] = HipKernels()   # raises without the .so or without a GPU
    return _default['kernels']


def default_device():
    return _default['device']


def set_default_kernels(kernels, device):
    """test hook: returns the previous (kernels, device) so it can be restored"""
    prev = (_default['kernels'], _default['device'])
    _default['kernels'], _default['device'] = kernels, device
    return prev
