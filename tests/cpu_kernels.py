"""
TEST DOUBLE (tests only): a torch-CPU emulation of every entry point behind
surreal_amd.kernels.HipKernels, with the same in/out tensor contract as the C ABI
(include/surreal_amd.h).  It exists so that the host logic of the product -- epoch control,
the device-side early-exit protocol, statistics slots, sharding + all-reduce across ranks --
can be exercised without a GPU (``-m "not gpu"`` tier and the world_size-2 gloo tests), and it
doubles as an executable statement of each kernel's contract for the GPU parity tests.

It is never importable from the product package and the product never selects it.
"""
import math

import numpy as np
import torch

from surreal_amd import _lib as L


def _f(x):
    return torch.as_tensor(x, dtype=torch.float32)


class TorchCpuKernels(object):
    name = 'torch-cpu-double'

    # ---- z-filter -----------------------------------------------------------------------
    def zfilter_stats(self, rs, rsq, cnt, eps, mean_out, std_out):
        mean = rs / cnt
        std = torch.clamp((rsq / cnt - mean.pow(2)).pow(0.5), min=eps)
        mean_out.copy_(mean)
        std_out.copy_(std)

    def zfilter_forward(self, x_view, mean, std, out):
        out.copy_(torch.clamp((x_view - mean) / std, -5.0, 5.0))

    def zfilter_forward_sums(self, x_view, rs, rsq, cnt, eps, out):
        mean = rs / cnt
        std = torch.clamp((rsq / cnt - mean.pow(2)).pow(0.5), min=eps)
        out.copy_(torch.clamp((x_view - mean) / std, -5.0, 5.0))

    def diaggauss_sample(self, mean, log_var, noise_scale, eps, actions, pd):
        A = mean.shape[1]
        std = torch.exp(log_var) * torch.ones_like(mean)
        if noise_scale is not None:
            std = std * noise_scale.view(-1, 1)
        act = mean if eps is None else eps * std + mean
        actions.copy_(torch.clamp(act, -1.0, 1.0))
        if pd is not None:
            pd[:, :A].copy_(mean)
            pd[:, A:].copy_(std)

    def zfilter_update_ws_floats(self, rows, D):
        return 0

    def zfilter_update(self, x_view, rs, rsq, cnt, count_rows, ws=None):
        rs += torch.sum(x_view, dim=0)
        rsq += torch.sum(x_view * x_view, dim=0)
        cnt += float(count_rows)

    # ---- MLP ----------------------------------------------------------------------------
    def layernorm_forward(self, x, gamma, beta, eps, y, mean=None, rstd=None):
        m = x.mean(1, keepdim=True)
        var = ((x - m) ** 2).mean(1, keepdim=True)
        rs = 1.0 / torch.sqrt(var + eps)
        y.copy_((x - m) * rs * gamma + beta)
        if mean is not None:
            mean.copy_(m.view(-1))
        if rstd is not None:
            rstd.copy_(rs.view(-1))

    def layernorm_backward_ws_floats(self, rows, F):
        return 1

    def layernorm_backward(self, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, ws, relu_mask=False):
        xh = (x - mean.view(-1, 1)) * rstd.view(-1, 1)
        g = dy * gamma
        v = rstd.view(-1, 1) * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
        if relu_mask:
            v = v * (x > 0)
        dgamma.copy_((dy * xh).sum(0))
        dbeta.copy_(dy.sum(0))
        dx.copy_(v)

    def mlp3_dgrad_rows_ws_floats(self, net):
        return 0

    def mlp3_packed_numel(self, net):
        return net.numel          # opaque to the caller

    def mlp3_pack(self, net, packed):
        o = 0
        for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3'):
            n = net.views[k].numel()
            packed[o:o + n].copy_(net.views[k].reshape(-1))
            o += n

    def mlp3_pack_zstats(self, net, packed, zf):
        self.mlp3_pack(net, packed)
        self.zfilter_stats(zf.running_sum, zf.running_sumsq, zf.count, zf.eps, zf._mean, zf._std)

    @staticmethod
    def _unpack(packed, net):
        out, o = {}, 0
        for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3'):
            n = net.views[k].numel()
            out[k] = packed[o:o + n].view(net.views[k].shape)
            o += n
        return out

    @staticmethod
    def _act(x, act):
        if act == L.SMX_ACT_RELU:
            return torch.relu(x)
        if act == L.SMX_ACT_TANH:
            return torch.tanh(x)
        return x

    def fused_exact_zfilter(self, on):
        pass                    # the double always divides

    def mlp3_forward_fused(self, packed, net, x_main, x_tail, zmean, zstd, out, act):
        x = x_main if x_tail is None else torch.cat([x_main, x_tail], dim=1)
        x = x.reshape(-1, x.shape[-1])
        if zmean is not None:
            x = torch.clamp((x - zmean) / zstd, -5.0, 5.0)
        w = self._unpack(packed, net)
        h = torch.relu(torch.nn.functional.linear(x, w['W1'], w['b1']))
        h = torch.relu(torch.nn.functional.linear(h, w['W2'], w['b2']))
        y = self._act(torch.nn.functional.linear(h, w['W3'], w['b3']), act)
        out.view(-1, net.OUT).copy_(y)

    def mlp3_forward(self, net, x, h1, h2, out, act, stop=None, pack=None):
        if stop is not None and int(stop[0]) != 0:
            return
        v = net.views
        h1.copy_(torch.relu(torch.nn.functional.linear(x, v['W1'], v['b1'])))
        h2.copy_(torch.relu(torch.nn.functional.linear(h1, v['W2'], v['b2'])))
        out.copy_(self._act(torch.nn.functional.linear(h2, v['W3'], v['b3']), act))

    def mlp3_forward_multi(self, jobs):
        for j in jobs:
            if j.get('stop') is not None and int(j['stop'][0]) != 0:
                continue
            self.mlp3_forward(j['net'], j['x'], j['h1'], j['h2'], j['out'], j.get('act', 0))
            if j.get('h1T') is not None:
                j['h1T'].copy_(j['h1'].t())
                j['h2T'].copy_(j['h2'].t())

    def mlp3_backward_multi(self, jobs):
        for j in jobs:
            if j.get('dz1T') is not None and j.get('xT') is not None:
                # the transposed operands must be what the row-major ones say (contract check)
                assert torch.equal(j['xT'], j['x'].t()) and torch.equal(j['h1T'], j['h1'].t())
                assert torch.equal(j['h2T'], j['h2'].t())
                assert torch.equal(j['dz3T'].reshape(j['dz3'].shape[1], -1), j['dz3'].t())
            self.mlp3_backward(j['net'], j['x'], j['h1'], j['h2'], j['dz3'], j['dz2'], j['dz1'],
                               j['grads'], j.get('sumsq'), j.get('stop'))
            if j.get('dz1T') is not None and not (j.get('stop') is not None and int(j['stop'][0]) != 0):
                j['dz2T'].copy_(j['dz2'].t())
                j['dz1T'].copy_(j['dz1'].t())

    def mlp3_backward_partials(self, net):
        t = lambda a: (a + 31) // 32  # noqa: E731
        return t(net.H1) * t(net.D) + t(net.H2) * t(net.H1) + t(net.OUT) * t(net.H2)

    def mlp3_backward_ws_floats(self, net, rows):
        return 0

    def mlp3_backward(self, net, x, h1, h2, dz3, dz2, dz1, grads, sumsq, stop=None, ws=None, packT=None, dx=None):
        if stop is not None and int(stop[0]) != 0:
            return False
        v = net.views
        dz2.copy_((dz3 @ v['W3']) * (h2 > 0))
        dz1.copy_((dz2 @ v['W2']) * (h1 > 0))
        pieces = [dz1.t() @ x, dz1.sum(0), dz2.t() @ h1, dz2.sum(0), dz3.t() @ h2, dz3.sum(0)]
        flat = torch.cat([p.reshape(-1) for p in pieces])
        grads[:flat.numel()].copy_(flat)
        if sumsq is not None:
            n = self.mlp3_backward_partials(net)
            sumsq[:n].zero_()
            sumsq[0] = float((flat.double() ** 2).sum())

    # ---- fused row-block epoch kernels ------------------------------------------------------
    def epoch_supported(self, *nets):
        return all(net.H1 % 4 == 0 and net.H2 % 4 == 0 and net.OUT <= 32 for net in nets)

    def epoch_blocks(self, rows):
        return (rows + 15) // 16

    def epoch_packed_numel(self, net):
        return net.numel            # opaque to the caller

    def epoch_pack(self, items):
        for net, packed in items:      # the double keeps a flat copy: the forward must see THESE values
            flat = torch.cat([net.views[k].reshape(-1) for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')])
            packed[:flat.numel()].copy_(flat)

    def epoch_prepare(self, obs0, xn, xnT, xr, zmean=None, zstd=None, ref_filter=None, obs_next=None, xnext=None,
                      ref_log_var=None, ref_std=None, pack=(), zero_words=None):
        if zmean is not None:
            self.zfilter_forward(obs0, zmean, zstd, xn)
        else:
            xn.copy_(obs0)
        if xnT is not None:
            xnT.copy_(xn.t())
        if ref_filter is not None:
            self.zfilter_forward_sums(obs0, ref_filter.running_sum, ref_filter.running_sumsq, ref_filter.count,
                                      ref_filter.eps, xr)
        else:
            xr.copy_(obs0)
        if xnext is not None:
            if zmean is not None:
                self.zfilter_forward(obs_next, zmean, zstd, xnext)
            else:
                xnext.copy_(obs_next)
        if ref_std is not None:
            ref_std.copy_(torch.exp(ref_log_var).view(1, -1).expand(ref_std.shape))
        if zero_words is not None:
            zero_words.zero_()
        if pack:
            self.epoch_pack(list(pack))

    @staticmethod
    def _packed_views(pk, net):
        W, o = {}, 0
        for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3'):
            n_ = net.views[k].numel()
            W[k] = pk[o:o + n_].view(net.views[k].shape)
            o += n_
        return W

    def epoch_forward(self, jobs, loss=None, ctrl=None, n_total=0):
        for j in jobs:
            if j.get('stop') is not None and int(j['stop'][0]) != 0:
                continue
            net, x = j['net'], j['x']
            v = net.views
            W = self._packed_views(j['packed'], net)   # weights from the packed copy, biases from the net
            h1 = torch.relu(torch.nn.functional.linear(x, W['W1'], v['b1']))
            h2 = torch.relu(torch.nn.functional.linear(h1, W['W2'], v['b2']))
            out = self._act(torch.nn.functional.linear(h2, W['W3'], v['b3']), j.get('act', 0))
            if j.get('h1T') is not None:
                j['h1T'].copy_(h1.t())
                j['h2T'].copy_(h2.t())
            if j.get('out') is not None:
                j['out'].copy_(out.view(j['out'].shape))
            if j.get('loss') == 'policy':
                self.policy_loss(loss['mode'], out, loss['log_var'], loss['actions'], loss['behave'],
                                 loss['ref'], loss['adv'], ctrl, loss['g_surr'], loss['g_kl'],
                                 loss['partials'])
            elif j.get('loss') == 'value':
                vv, g = out.view(-1), loss['returns'].view(-1)
                rows = vv.numel()
                loss['v_dz3'].view(-1).copy_(2.0 * (vv - g) / float(n_total))
                d = g - vv
                for b in range(self.epoch_blocks(rows)):
                    sl = slice(16 * b, min(16 * (b + 1), rows))
                    db, gb = d[sl].double(), g[sl].double()
                    loss['v_partials'][b] = _f([db.numel(), db.mean(), ((db - db.mean()) ** 2).sum(), gb.mean(),
                                                ((gb - gb.mean()) ** 2).sum(), (db ** 2).sum(), 0, 0])
                if loss.get('v_will_update'):
                    ctrl.view(torch.int32)[L.C_STEP_CRITIC] += 1

    def epoch_backward(self, jobs, loss, ctrl, n_total):
        for j in jobs:
            net = j['net']
            v = net.views
            rows = j['x'].shape[0]
            if j.get('loss') == 'policy':
                ci = ctrl.view(torch.int32)
                if int(ci[L.C_STOP]) != 0:
                    continue
                A = net.OUT
                dz3 = torch.zeros(rows, A)
                self.policy_finalize(loss['mode'], loss['partials'], self.epoch_blocks(rows), loss['g_surr'],
                                     loss['g_kl'], loss['log_var'], n_total, ctrl, loss['check_stop'],
                                     loss['will_update'], dz3, loss['dlogvar'], loss.get('dlogvar_sumsq'),
                                     loss['stats'])
                if int(ci[L.C_STOP]) != 0 or not loss['will_update']:
                    continue
                j['dz3T'].copy_(dz3.t())
            elif j.get('loss') in ('rhs_surr', 'rhs_kl'):
                if int(ctrl.view(torch.int32)[L.C_STOP]) != 0:
                    continue
                dz3 = (loss['g_surr'] if j['loss'] == 'rhs_surr' else loss['g_kl']) / float(n_total)
                j['dz3T'].copy_(dz3.t())
            else:
                dz3 = j['dz3'].view(rows, 1)
            h2, h1 = j['h2T'].t(), j['h1T'].t()
            W = self._packed_views(j['packed'], net)
            dz2 = (dz3 @ W['W3']) * (h2 > 0)
            dz1 = (dz2 @ W['W2']) * (h1 > 0)
            j['dz2T'].copy_(dz2.t())
            j['dz1T'].copy_(dz1.t())

    def epoch_fwdbwd_supported(self, *nets):
        return self.epoch_supported(*nets)

    def epoch_fwdbwd(self, jobs, loss, ctrl, n_total, sync_word, kl_slots=None):
        """the contract of smx_epoch_fwdbwd_f32: the two launches it replaces, back to back"""
        self.epoch_forward(jobs, loss, ctrl, n_total)
        self.epoch_backward(jobs, loss, ctrl, n_total)
        if any(j.get('loss') == 'policy' for j in jobs):
            sync_word += self.epoch_blocks(jobs[0]['x'].shape[0])

    def mlp3_wgrad_multi(self, jobs):
        for j in jobs:
            if j.get('stop') is not None and int(j['stop'][0]) != 0:
                continue
            net = j['net']
            dz1, dz2 = j['dz1T'].t(), j['dz2T'].t()
            dz3 = j['dz3T'].reshape(net.OUT, -1)[:, :dz1.shape[0]].t()
            x, h1, h2 = j['xT'].t(), j['h1T'].t(), j['h2T'].t()
            pieces = [dz1.t() @ x, dz1.sum(0), dz2.t() @ h1, dz2.sum(0), dz3.t() @ h2, dz3.sum(0)]
            flat = torch.cat([p.reshape(-1) for p in pieces])
            j['grads'][:flat.numel()].copy_(flat)
            if j.get('sumsq') is not None:
                n = self.mlp3_backward_partials(net)
                j['sumsq'][:n].zero_()
                j['sumsq'][0] = float((flat.double() ** 2).sum())

    # ---- GAE / normalisation ------------------------------------------------------------
    def gae(self, values, rewards, dones, gpow, lpow, gamma, gamma_H, B, N, H, adv, ret,
            values_tail=None):
        if values_tail is not None:
            v = torch.cat([values.view(B, N), values_tail.view(B, 1)], 1)
        else:
            v = values.view(B, N + 1).clone()
        v[:, 1:] *= 1 - dones
        tds = rewards + gamma * v[:, 1:] - v[:, :-1]
        E = N - H + 1
        g, l = gpow[:H], lpow[:H]
        r_out, a_out = torch.zeros(B, E), torch.zeros(B, E)
        for s in range(E):
            r_out[:, s] = torch.sum(g * rewards[:, s:s + H], 1) + v[:, s + H] * gamma_H
            a_out[:, s] = torch.sum(tds[:, s:s + H] * g * l, 1)
        adv.view(B, E).copy_(a_out)
        ret.view(B, E).copy_(r_out)

    def gae_norm(self, values, rewards, dones, gpow, lpow, gamma, gamma_H, B, N, H, adv, ret, adv_mom, min_std, ticket,
                 values_tail=None):
        self.gae(values, rewards, dones, gpow, lpow, gamma, gamma_H, B, N, H, adv, ret, values_tail=values_tail)
        self.moments(adv, adv_mom)
        self.adv_normalize(adv, adv_mom, min_std)

    def reward_filter_partials(self):
        return 128

    def linear_multi(self, jobs):
        for j in jobs:
            kw = j[-1] if isinstance(j[-1], dict) else {}
            if j[0] == 'wgrad':
                self.linear_wgrad(*j[1:8], **kw)
            else:
                self.linear(*j[1:10], **kw)

    def frame_stack(self, frames, n_stack, start, n_step, stride, W, dst, episode_first=None):
        actors, R = frames.shape[:2]
        fr = frames.reshape(actors, R, -1)
        out = dst.view(actors * W, n_step, n_stack, -1)
        for a in range(actors):
            for w in range(W):
                for j in range(n_step):
                    s_ = start + w * stride + j
                    lo = int(episode_first[a, s_]) if episode_first is not None else 0
                    for i in range(n_stack):
                        out[a * W + w, j, i] = fr[a, max(s_ - (n_stack - 1) + i, lo)]

    def synth_frames(self, s0, t, dst):
        n, C, H, W = dst.shape
        c, y, x = np.meshgrid(np.arange(C), np.arange(H), np.arange(W), indexing='ij')
        base = torch.as_tensor(37 * c + 5 * y + 11 * x)
        for a in range(n):
            shift = 3 * int(t) + int(100 * abs(float(s0[a])))
            dst[a] = ((base + shift) % 256).to(torch.uint8)

    def synth_rollout_supported(self, net):
        return False                 # the double walks the rollout step by step (SyntheticVecEnv's layered path)

    def reward_filter(self, rewards, scale, state, eps, out, partials, ticket, use_filter=True, update=True, sums=None):
        x = rewards * _f(scale)
        if use_filter:
            mean = state[1] / state[0]
            std = torch.clamp((state[2] / state[0] - mean.pow(2)).pow(0.5), min=eps)
            out.copy_(torch.clamp((x - mean) / std, -5.0, 5.0))
        else:
            out.copy_(x)
        s1, s2 = x.double().sum(), (x.double() * x.double()).sum()
        if sums is not None:
            sums.copy_(torch.stack([torch.tensor(float(x.numel())), s1.float(), s2.float()]))
        if update:
            state[0] += float(x.numel())
            state[1] += s1.float()
            state[2] = s2.float()

    def learn_epilogue(self, ret, ret_mom, log_var, out4, ticket, zfilter=None, x=None, count_rows=0, v_partials=None,
                       n_epochs=0, nblk=0, v_stats=None, stats_stride=0):
        if n_epochs:
            self.value_finalize(v_partials, n_epochs, nblk, v_stats, stats_stride)
        self.moments(ret, ret_mom)
        if zfilter is not None:
            self.zfilter_update(x, zfilter.running_sum, zfilter.running_sumsq, zfilter.count, count_rows)
        self.final_stats(log_var, zfilter, out4)

    def moments(self, x, out):
        xd = x.double().reshape(-1)
        mean = xd.mean()
        out.copy_(_f([xd.numel(), float(mean), float(((xd - mean) ** 2).sum())]))

    def moments_merge(self, parts, out):
        n = mean = m2 = 0.0
        for nb, mb, qb in parts.view(-1, 3).double().tolist():
            if nb <= 0:
                continue
            nt, d = n + nb, mb - mean
            m2 += qb + d * d * n * nb / nt
            mean += d * nb / nt
            n = nt
        out.copy_(_f([n, mean, m2]))

    def adv_normalize(self, x, mom, min_std):
        n, mean, m2 = mom[0], mom[1], mom[2]
        std = torch.sqrt(m2 / (n - 1.0))
        den = _f(min_std) if min_std > float(std) else std
        x.copy_((x - mean) / den)

    # ---- losses -------------------------------------------------------------------------
    LOSS_ROWS = 16           # rows per workgroup of the policy loss kernel (partial-sum granularity)

    def loss_blocks(self, rows):
        return (rows + self.LOSS_ROWS - 1) // self.LOSS_ROWS

    def policy_loss(self, mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl,
                    partials):
        ci = ctrl.view(torch.int32)
        if int(ci[L.C_STOP]) != 0:
            return
        rows, A = mean.shape
        sig = torch.exp(log_var).view(1, A)
        mb, sb = behave[:, :A], behave[:, A:]
        mr, sr = ref[:, :A], ref[:, A:]
        c = 0.5 * np.log(2.0 * np.pi) * A

        def ll(mu, s):
            return -0.5 * (((actions - mu) / s) ** 2).sum(1) - c - torch.log(s).expand(rows, A).sum(1)

        el = torch.exp(ll(mean, sig))
        Ll = torch.clamp(el, min=1e-5)
        Lb = torch.clamp(torch.exp(ll(mb, sb)), min=1e-5)
        kl = torch.log(sig / sr).sum(1) + ((sr ** 2 + (mr - mean) ** 2) / (2.0 * sig ** 2)).sum(1) - 0.5 * A
        klb = torch.log(sb / sr).sum(1) + ((sr ** 2 + (mr - mb) ** 2) / (2.0 * sb ** 2)).sum(1) - 0.5 * A
        ad = adv.view(-1)
        if mode == L.SMX_PPO_CLIP:
            eps = float(ctrl[L.C_CLIP_EPS])
            ratio = Ll / Lb
            cr = torch.clamp(ratio, 1 - eps, 1 + eps)
            surr, cs = -ratio * ad, -cr * ad
            loss_r = torch.maximum(surr, cs)
            dLl = torch.where(surr >= cs, -ad / Lb, torch.zeros_like(ad))
        else:
            Lbc = torch.clamp(Lb, min=1e-2)
            surr = -(ad * (Ll / Lbc))
            loss_r = surr
            dLl = -ad / Lbc
        dll = torch.where(el >= 1e-5, dLl * el, torch.zeros_like(el))
        dt = 1.0 - mean ** 2
        g_surr.copy_(dll.view(-1, 1) * ((actions - mean) / sig ** 2) * dt)
        g_kl.copy_(((mean - mr) / sig ** 2) * dt)
        z2 = ((actions - mean) / sig) ** 2
        gs = dll.view(-1, 1) * (z2 - 1.0)
        gk = 1.0 - (sr ** 2 + (mr - mean) ** 2) / sig ** 2
        isw = Ll / (Lb + 1e-4)
        nblk = self.loss_blocks(rows)
        partials[:nblk].zero_()       # like the kernels: a launch rewrites the rows of ITS blocks, nothing else
        for b in range(nblk):
            sl = slice(self.LOSS_ROWS * b, min(self.LOSS_ROWS * (b + 1), rows))
            partials[b, 0] = surr[sl].sum()
            partials[b, 1] = loss_r[sl].sum()
            partials[b, 2] = kl[sl].sum()
            partials[b, 3] = Lb[sl].sum()
            partials[b, 4] = isw[sl].sum()
            partials[b, 5] = klb[sl].sum()
            partials[b, 8:8 + A] = gs[sl].sum(0)
            partials[b, 8 + A:8 + 2 * A] = gk[sl].sum(0)

    def partials_fold(self, partials, nblk, out, ctrl=None):
        if ctrl is not None and int(ctrl.view(torch.int32)[L.C_STOP]) != 0:
            return
        nout = out.shape[0]
        R = (nblk + nout - 1) // nout
        for j in range(nout):
            out[j] = partials[j * R:min(nblk, (j + 1) * R)].sum(0)

    def policy_finalize(self, mode, partials, nblk, g_surr, g_kl, log_var, n_total, ctrl,
                        check_stop, will_update, dz3, dlogvar, dlogvar_sumsq, stats, dz3_t=None):
        ci = ctrl.view(torch.int32)
        if int(ci[L.C_STOP]) != 0:
            return
        rows, A = g_surr.shape
        S = partials[:nblk].sum(0)
        n = float(n_total)
        surr_mean, kl_mean = S[0] / n, S[2] / n
        c_kl = 0.0
        if mode == L.SMX_PPO_CLIP:
            loss = S[1] / n
        else:
            beta, eta, kt = float(ctrl[L.C_BETA]), float(ctrl[L.C_ETA]), float(ctrl[L.C_KL_TARGET])
            loss = surr_mean + beta * kl_mean
            c_kl = beta
            if float(kl_mean) - 2.0 * kt > 0:
                d = kl_mean - _f(2.0 * kt)
                loss = loss + eta * (d * d)
                c_kl = c_kl + 2.0 * eta * float(d)
        dz3.copy_((g_surr + c_kl * g_kl) / n)
        if dz3_t is not None:
            dz3_t.copy_(dz3.t())
        gl = (S[8:8 + A] + c_kl * S[8 + A:8 + 2 * A]) / n
        dlogvar.copy_(gl)
        if dlogvar_sumsq is not None:
            dlogvar_sumsq.copy_((gl * gl).sum().view(1))
        stats[L.PS_SURR] = surr_mean
        stats[L.PS_LOSS] = loss
        stats[L.PS_ENTROPY] = 0.5 * torch.log(torch.exp(log_var)).sum() + .5 * np.log(2 * np.pi * np.e) * A
        stats[L.PS_KL] = kl_mean
        stats[L.PS_LB] = S[3] / n
        stats[L.PS_ISW] = S[4] / n
        stats[L.PS_REFBEH] = S[5] / n
        if check_stop and float(kl_mean) > 4.0 * float(ctrl[L.C_KL_TARGET]):
            ci[L.C_STOP] = 1
        elif will_update:
            ci[L.C_STEP_ACTOR] += 1
            ci[L.C_EPOCHS_DONE] += 1

    def epoch_losses(self, mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl,
                     partials, check_stop, will_update, dz3, dlogvar, dlogvar_sumsq, stats, dz3_t=None,
                     values=None, returns=None, v_dz3=None, v_partials=None, v_will_update=True):
        rows = mean.shape[0]
        # the value blocks do not look at the stop flag; the policy part is skipped when it is set
        if values is not None:
            self.value_loss(values, returns, rows, v_dz3, v_partials, ctrl, v_will_update)
        self.policy_loss(mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl, partials)
        self.policy_finalize(mode, partials, self.loss_blocks(rows), g_surr, g_kl, log_var, rows, ctrl,
                             check_stop, will_update, dz3, dlogvar, dlogvar_sumsq, stats, dz3_t=dz3_t)

    def epoch_losses_dp(self, mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl,
                        partials, n_total, g_surr_t=None, g_kl_t=None, values=None, returns=None,
                        v_dz3=None, v_partials=None, v_will_update=True):
        if values is not None:
            self.value_loss(values, returns, n_total, v_dz3, v_partials, ctrl, v_will_update)
        if int(ctrl.view(torch.int32)[L.C_STOP]) != 0:
            return
        self.policy_loss(mode, mean, log_var, actions, behave, ref, adv, ctrl, g_surr, g_kl, partials)
        g_surr.mul_(_f(1.0) / _f(float(n_total)))
        g_kl.mul_(_f(1.0) / _f(float(n_total)))
        if g_surr_t is not None:
            g_surr_t.copy_(g_surr.t())
            if g_kl_t is not None:
                g_kl_t.copy_(g_kl.t())

    def epoch_combine(self, mode, partials, nblk, n_total, log_var, ctrl, check_stop, will_update,
                      stats, grads_a, grads_kl, n_mlp, sumsq_a, grads_c=None, sumsq_c=None):
        if grads_c is not None:
            self.sumsq_partials(grads_c, sumsq_c)
        if int(ctrl.view(torch.int32)[L.C_STOP]) != 0:
            return
        A = log_var.numel()
        # the scalar part of policy_finalize on zero rows of gradient tiles; its c_kl comes back
        # through log_var's gradient (dlogvar = (S_ll + c_kl * S_kl) / n)
        empty = torch.zeros(0, A)
        dlv = torch.zeros(A)
        beta, eta, kt = float(ctrl[L.C_BETA]), float(ctrl[L.C_ETA]), float(ctrl[L.C_KL_TARGET])
        S = partials[:nblk].sum(0)
        kl_mean = S[2] / float(n_total)
        c_kl = 0.0
        if mode != L.SMX_PPO_CLIP:
            c_kl = beta
            if float(kl_mean) - 2.0 * kt > 0:
                c_kl = c_kl + 2.0 * eta * float(kl_mean - _f(2.0 * kt))
        self.policy_finalize(mode, partials, nblk, empty, empty, log_var, n_total, ctrl, check_stop,
                             will_update, torch.zeros(0, A), dlv, None, stats)
        if int(ctrl.view(torch.int32)[L.C_STOP]) != 0:
            sumsq_a.zero_()              # this epoch took the early exit: no gradient is formed
            return
        if grads_kl is not None:
            grads_a[:n_mlp].add_(c_kl * grads_kl[:n_mlp])
        grads_a[n_mlp:n_mlp + A].copy_(dlv)
        self.sumsq_partials(grads_a, sumsq_a)

    def final_stats(self, log_var, zfilter, out4):
        out4.zero_()
        out4[0] = log_var.double().mean()
        if zfilter is not None:
            m = zfilter.running_sum / zfilter.count
            q = zfilter.running_sumsq / zfilter.count
            out4[1], out4[2] = m.double().mean(), q.double().mean()
            out4[3] = (q - m * m).pow(0.5).double().mean()

    def value_loss_blocks(self, rows):
        return (rows + 255) // 256

    def value_loss(self, values, returns, n_total, dz3, partials, ctrl, will_update):
        v, g = values.view(-1), returns.view(-1)
        rows = v.numel()
        dz3.view(-1).copy_(2.0 * (v - g) / float(n_total))
        d = g - v
        for b in range(self.value_loss_blocks(rows)):
            sl = slice(256 * b, min(256 * (b + 1), rows))
            db, gb = d[sl].double(), g[sl].double()
            partials[b] = _f([db.numel(), db.mean(), ((db - db.mean()) ** 2).sum(), gb.mean(),
                              ((gb - gb.mean()) ** 2).sum(), (db ** 2).sum(), 0, 0])
        if will_update:
            ctrl.view(torch.int32)[L.C_STEP_CRITIC] += 1

    def value_finalize(self, partials, count, nblk, stats, stride):
        for e in range(count):
            n = md = qd = mg = qg = sq = 0.0
            for row in partials[e, :nblk].double().tolist():
                nb = row[0]
                if nb <= 0:
                    continue
                nt = n + nb
                dl = row[1] - md
                qd += row[2] + dl * dl * n * nb / nt
                md += dl * nb / nt
                dl = row[3] - mg
                qg += row[4] + dl * dl * n * nb / nt
                mg += dl * nb / nt
                sq += row[5]
                n = nt
            stats[e, L.VS_LOSS] = sq / n
            stats[e, L.VS_EXPVAR] = 1.0 - (qd / (n - 1.0)) / (qg / (n - 1.0)) if n > 1 else float('nan')

    # ---- optimiser ----------------------------------------------------------------------
    def clip_adam(self, theta, grads, m, v, sumsq, npart, ctrl, which, honour_stop, grad_norm_out, pack=None):
        self._clip_adam(theta, grads, m, v, sumsq, npart, ctrl, which, honour_stop, grad_norm_out)
        if pack is not None:
            self.epoch_pack([pack])

    def _clip_adam(self, theta, grads, m, v, sumsq, npart, ctrl, which, honour_stop, grad_norm_out):
        ci = ctrl.view(torch.int32)
        if honour_stop and int(ci[L.C_STOP]) != 0:
            return
        norm = torch.sqrt(sumsq[:npart].sum())
        max_norm = float(ctrl[L.C_CRITIC_MAX_NORM if which else L.C_ACTOR_MAX_NORM])
        coef = 1.0
        if max_norm > 0:
            coef = min(max_norm / (float(norm) + 1e-6), 1.0)
        if grad_norm_out is not None:
            grad_norm_out.copy_(norm.view(1))
        step = int(ci[L.C_STEP_CRITIC if which else L.C_STEP_ACTOR])
        lr = float(ctrl[L.C_LR_CRITIC if which else L.C_LR_ACTOR])
        wd = float(ctrl[L.C_CRITIC_WD if which else L.C_ACTOR_WD])
        b1, b2, eps = 0.9, 0.999, 1e-8
        g = grads * coef
        if wd != 0:
            g = g + wd * theta
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        theta.addcdiv_(m, denom, value=-(lr / bc1))

    def clip_adam_pair(self, actor, critic, ctrl, pack=None):
        for which, (theta, grads, m, v, sumsq, npart, honour_stop, gno) in enumerate((actor, critic)):
            self.clip_adam(theta, grads, m, v, sumsq, npart, ctrl, which, honour_stop, gno)
        if pack is not None:
            self.epoch_pack(list(pack))

    def sumsq_blocks(self, n):
        return max(1, min(256, (n + 4095) // 4096))

    def sumsq_partials(self, x, partials):
        nb = self.sumsq_blocks(x.numel())
        partials[:nb].zero_()
        partials[0] = (x.double() ** 2).sum()

    # ---- replay / windowing ---------------------------------------------------------------
    def ring_insert(self, table, cursor, src):
        cap = table.shape[0]
        idx = (int(cursor) + torch.arange(src.shape[0])) % cap
        table[idx] = src

    def gather_rows(self, table, idx, dst):
        dst.copy_(table[idx.clamp(0, table.shape[0] - 1)])

    def uniform_indices(self, idx, length, seed, offset):
        g = torch.Generator().manual_seed((int(seed) * 1000003 + int(offset)) % (2 ** 63))
        idx.copy_(torch.randint(0, int(length), idx.shape, generator=g))

    def uniform_gather_multi(self, tables, outs, length, seed, offset, idx=None, idx_out=None):
        if idx is None:
            idx = torch.empty(outs[0].shape[0], dtype=torch.int64)
            self.uniform_indices(idx, length, seed, offset)
        if idx_out is not None:
            idx_out.copy_(idx)
        j = idx.clamp(0, tables[0].shape[0] - 1)
        for t, o in zip(tables, outs):
            o.copy_(t[j])

    def window_emit(self, src, start, n_step, stride, W, dst):
        actors, T, width = src.shape
        out = torch.stack([src[:, start + w * stride:start + w * stride + n_step] for w in range(W)], 1)
        dst.view(actors, W, n_step, width).copy_(out)

    def synth_act_env_step_head(self, W3, b3, h2, out_act, state, init_state, log_var, noise_scale, eps, t,
                                episode_len, slot, rolls, zfilter, xn_out):
        mean = self._act(torch.nn.functional.linear(h2, W3, b3), out_act)
        self.synth_act_env_step(state, init_state, mean, log_var, noise_scale, eps, t, episode_len, slot, rolls,
                                zfilter, xn_out)

    def synth_env_step(self, state, init_state, actions, t, episode_len, slot, obs_roll, act_roll,
                       rew_roll, done_roll):
        n, D = state.shape
        A = actions.shape[1]
        ac = actions.clamp(-1.0, 1.0)
        k = torch.arange(D)
        drift = 0.01 * (((37 * k) % 17) - 8).float()
        sn = ((0.9 * state + 0.5 * ac[:, k % A]) + drift).clamp(-10.0, 10.0)
        done = (t + 1 >= episode_len)
        if obs_roll is not None:
            obs_roll[:, slot] = state
            if slot + 1 < obs_roll.shape[1]:
                obs_roll[:, slot + 1] = sn
            act_roll[:, slot] = ac
            rew_roll[:, slot] = (-0.1 * (ac.double() ** 2).sum(1) + 0.05 * sn[:, 0].double()).float()
            done_roll[:, slot] = 1.0 if done else 0.0
        state.copy_(init_state if done else sn)

    def synth_act_env_step(self, state, init_state, mean, log_var, noise_scale, eps, t, episode_len,
                           slot, rolls, zfilter, xn_out):
        n, A = mean.shape
        r = rolls or {}
        acts = torch.empty(n, A)
        pd = torch.empty(n, 2 * A)
        self.diaggauss_sample(mean, log_var, noise_scale, eps, acts, pd)
        if 'pds' in r:
            r['pds'][:, slot] = pd
        self.synth_env_step(state, init_state, acts, t, episode_len, slot, r.get('obs'), r.get('actions'),
                            r.get('rewards'), r.get('dones'))
        if xn_out is not None:
            if zfilter is not None:
                self.zfilter_forward_sums(state, zfilter.running_sum, zfilter.running_sumsq, zfilter.count,
                                          zfilter.eps, xn_out)
            else:
                xn_out.copy_(state)

    # ---- generic dense layer + DDPG pieces ---------------------------------------------------
    def linear(self, A, a_kc, B, b_kc, bias, C, M, N, K, act=0, relu_mask=None, lda=None, ldb=None,
               ldc=None, stop=None):
        if stop is not None and int(stop[0]) != 0:
            return
        lda = lda if lda is not None else A.stride(0)
        ldb = ldb if ldb is not None else B.stride(0)
        ldc = ldc if ldc is not None else C.stride(0)

        def mat(X, ld, kc, rows):   # -> [rows, K]
            flat = X.reshape(-1) if X.is_contiguous() else None
            base = torch.as_strided(X, (rows, K), (ld, 1) if kc else (1, ld))
            return base
        a, b = mat(A, lda, a_kc, M), mat(B, ldb, b_kc, N)
        out = a @ b.t()
        if bias is not None:
            out = out + bias[:N]
        out = self._act(out, act)
        if relu_mask is not None:
            out = out * (torch.as_strided(relu_mask, (M, N), (ldc, 1)) > 0)
        torch.as_strided(C, (M, N), (ldc, 1)).copy_(out)

    def linear_wgrad(self, dZ, X, dW, db, M, N, rows, ldz=None, ldx=None, ldw=None, ws=None):
        ldz = ldz if ldz is not None else dZ.stride(0)
        ldx = ldx if ldx is not None else X.stride(0)
        ldw = ldw if ldw is not None else dW.stride(0)
        dz = torch.as_strided(dZ, (rows, M), (ldz, 1))
        x = torch.as_strided(X, (rows, N), (ldx, 1))
        torch.as_strided(dW, (M, N), (ldw, 1)).copy_(dz.t() @ x)
        if db is not None:
            db.view(-1)[:M].copy_(dz.sum(0))

    # ---- LSTM stem (contract of smx_lstm_forward_f32 / smx_lstm_backward_f32) ------------------
    def lstm_forward(self, net, x, B, T, h0, c0, gates, out, cs, hprev=None, hN=None, cN=None,
                     stop=None):
        if stop is not None and int(stop[0]) != 0:
            return
        v, H = net.views, net.H
        xs = x.reshape(B, T, -1)
        h = h0.reshape(B, H).clone() if h0 is not None else torch.zeros(B, H)
        c = c0.reshape(B, H).clone() if c0 is not None else torch.zeros(B, H)
        G, O, C = gates.view(B, T, 4 * H), out.view(B, T, H), cs.view(B, T, H)
        for t in range(T):
            g = (xs[:, t] @ v['weight_ih'].t() + v['bias_ih']) + (h @ v['weight_hh'].t() + v['bias_hh'])
            i, f, gg, o = g.chunk(4, 1)
            i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
            if hprev is not None:
                hprev.view(B, T, H)[:, t] = h
            c = f * c + i * gg
            h = o * torch.tanh(c)
            G[:, t] = torch.cat([i, f, gg, o], 1)
            O[:, t] = h
            C[:, t] = c
        if hN is not None:
            hN.copy_(h)
        if cN is not None:
            cN.copy_(c)

    def linear_wgrad_ws_floats(self, M, N, rows):
        return 0

    def lstm_backward_ws_floats(self, net, B, T):
        return 0

    def lstm_backward(self, net, x, B, T, c0, gates, cs, hprev, dout, dgates, grads, stop=None,
                      ws=None):
        if stop is not None and int(stop[0]) != 0:
            return
        v, H = net.views, net.H
        G = gates.view(B, T, 4 * H).clone()          # dgates may alias gates
        C, DO = cs.view(B, T, H), dout.reshape(B, T, H)
        DG = dgates.view(B, T, 4 * H)
        dh_rec = torch.zeros(B, H)
        dc_rec = torch.zeros(B, H)
        for t in range(T - 1, -1, -1):
            i, f, gg, o = G[:, t].chunk(4, 1)
            c = C[:, t]
            cp = C[:, t - 1] if t > 0 else (c0.reshape(B, H) if c0 is not None else torch.zeros(B, H))
            dh = DO[:, t] + dh_rec
            tc = torch.tanh(c)
            dc = dc_rec + dh * o * (1 - tc * tc)
            d = torch.cat([dc * gg * i * (1 - i), dc * cp * f * (1 - f), dc * i * (1 - gg * gg),
                           dh * tc * o * (1 - o)], 1)
            DG[:, t] = d
            dc_rec = dc * f
            dh_rec = d @ v['weight_hh']
        d2 = DG.reshape(B * T, 4 * H)
        pieces = [d2.t() @ x.reshape(B * T, -1), d2.t() @ hprev.reshape(B * T, H), d2.sum(0), d2.sum(0)]
        grads.copy_(torch.cat([p.reshape(-1) for p in pieces]))

    # ---- CNN stem data movement (contracts of smx_im2col_f32 / smx_col2im_f32 / flatten_order) ---
    def im2col(self, src, F, C, Hin, Win, k, stride, cols, channel_last=False, scale_div=0.0):
        # buffers may be longer than F frames (a tail chunk of a workspace): only F frames are touched
        x = src.reshape(-1)[:F * C * Hin * Win].to(torch.float32)
        if channel_last:
            x = x.reshape(F, Hin, Win, C).permute(0, 3, 1, 2)
        x = x.reshape(F, C, Hin, Win)
        if scale_div:
            x = x / scale_div
        u = torch.nn.functional.unfold(x, k, stride=stride)          # [F, C*k*k, P]
        P = u.shape[2]
        cols[:F * P].copy_(u.transpose(1, 2).reshape(F * P, C * k * k))

    @staticmethod
    def conv_u8_supported(frames, C, Hin, Win, k, stride, cout, W=None):
        K = C * k * k
        return frames.dtype == torch.uint8 and cout <= 16 and k % 4 == 0 and Win % 4 == 0 and stride % 4 == 0 \
            and K % 64 == 0 and K <= 256

    def conv_u8_forward(self, frames, F, C, Hin, Win, k, stride, W, bias, cout, y, stop=None):
        if stop is not None and int(stop[0]) != 0:
            return
        Ho, Wo = (Hin - k) // stride + 1, (Win - k) // stride + 1
        cols = torch.empty(F * Ho * Wo, C * k * k)
        self.im2col(frames, F, C, Hin, Win, k, stride, cols, scale_div=255.0)
        y[:F * Ho * Wo].copy_(torch.relu(torch.nn.functional.linear(cols, W.reshape(cout, -1), bias)))

    def conv_u8_wgrad_ws_floats(self, cout, K):
        return 1

    def conv_u8_wgrad(self, frames, F, C, Hin, Win, k, stride, dy, cout, dW, db, ws, stop=None):
        if stop is not None and int(stop[0]) != 0:
            return
        Ho, Wo = (Hin - k) // stride + 1, (Win - k) // stride + 1
        rows = F * Ho * Wo
        cols = torch.empty(rows, C * k * k)
        self.im2col(frames, F, C, Hin, Win, k, stride, cols, scale_div=255.0)
        dW.view(cout, -1).copy_(dy[:rows].reshape(rows, cout).t() @ cols)
        if db is not None:
            db.view(-1)[:cout].copy_(dy[:rows].reshape(rows, cout).sum(0))

    @staticmethod
    def conv_cl_supported(src, C, k, cout):
        return src.dtype == torch.float32 and C == 16 and cout <= 32 and k in (2, 3, 4)

    def conv_cl_forward(self, src, F, C, Hin, Win, k, stride, W, bias, cout, y, stop=None):
        if stop is not None and int(stop[0]) != 0:
            return
        Ho, Wo = (Hin - k) // stride + 1, (Win - k) // stride + 1
        cols = torch.empty(F * Ho * Wo, C * k * k)
        self.im2col(src, F, C, Hin, Win, k, stride, cols, channel_last=True)
        y[:F * Ho * Wo].copy_(torch.relu(torch.nn.functional.linear(cols, W.reshape(cout, -1), bias)))

    def conv_cl_wgrad_ws_floats(self, cout, k):
        return 1

    def conv_cl_wgrad(self, src, F, C, Hin, Win, k, stride, dy, cout, dW, db, ws, stop=None):
        if stop is not None and int(stop[0]) != 0:
            return
        Ho, Wo = (Hin - k) // stride + 1, (Win - k) // stride + 1
        rows = F * Ho * Wo
        cols = torch.empty(rows, C * k * k)
        self.im2col(src, F, C, Hin, Win, k, stride, cols, channel_last=True)
        dW.view(cout, -1).copy_(dy[:rows].reshape(rows, cout).t() @ cols)
        if db is not None:
            db.view(-1)[:cout].copy_(dy[:rows].reshape(rows, cout).sum(0))

    @staticmethod
    def conv_cl_dgrad_supported(dy, C, k, stride, cout):
        return C == 16 and cout <= 32 and k == 2 * stride

    def conv_cl_dgrad(self, dy, F, C, Hin, Win, k, stride, W, cout, relu_of, dx, stop=None):
        if stop is not None and int(stop[0]) != 0:
            return
        Ho, Wo = (Hin - k) // stride + 1, (Win - k) // stride + 1
        rows = F * Ho * Wo
        dcols = dy[:rows].reshape(rows, cout) @ W.reshape(cout, -1)
        self.col2im(dcols, F, C, Hin, Win, k, stride, relu_of, dx)

    def col2im(self, dcols, F, C, Hin, Win, k, stride, relu_of, dx):
        P = ((Hin - k) // stride + 1) * ((Win - k) // stride + 1)
        u = dcols[:F * P].reshape(F, P, C * k * k).transpose(1, 2)
        x = torch.nn.functional.fold(u, (Hin, Win), k, stride=stride)       # [F, C, Hin, Win]
        x = x.permute(0, 2, 3, 1).reshape(F * Hin * Win, C)
        if relu_of is not None:
            x = x * (relu_of[:F * Hin * Win].reshape(F * Hin * Win, C) > 0)
        dx[:F * Hin * Win].copy_(x)

    def flatten_order(self, src, O, C, P, to_channel_last, out):
        if to_channel_last:
            out.copy_(src.reshape(O, C, P).transpose(1, 2).reshape(out.shape))
        else:
            out.copy_(src.reshape(O, P, C).transpose(1, 2).reshape(out.shape))

    def ddpg_critic_loss(self, q, q_next, rewards, dones, gamma_n, y, dz3):
        yy = rewards.view(-1) + gamma_n * q_next.view(-1) * (1.0 - dones.view(-1))
        y.view(-1).copy_(yy)
        dz3.view(-1).copy_(2.0 * (q.view(-1) - yy) / q.numel())

    def tanh_backward(self, da, a, out):
        assert da.is_contiguous() and a.is_contiguous() and out.is_contiguous()     # the kernel indexes linearly
        out.copy_(da * (1.0 - a * a))

    def fill(self, x, value):
        x.fill_(value)

    def adam_step(self, theta, grads, m, v, lr, step, weight_decay=0.0, clip_value=0.0):
        g = grads.clamp(-clip_value, clip_value) if clip_value > 0 else grads.clone()
        if weight_decay != 0:
            g = g + weight_decay * theta
        b1, b2, eps = 0.9, 0.999, 1e-8
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        theta.addcdiv_(m, denom, value=-(lr / bc1))

    def soft_update(self, target, source, tau):
        if tau >= 1.0:
            target.copy_(source)
        else:
            target.copy_(target * (1.0 - tau) + source * tau)

    def ddpg_critic_loss_step(self, q, q_next, rewards, dones, gamma_n, y, dz3, step):
        self.ddpg_critic_loss(q, q_next, rewards, dones, gamma_n, y, dz3)
        step += 1

    def adam_step_dev(self, theta, grads, m, v, lr, step, weight_decay=0.0, clip_value=0.0):
        self.adam_step(theta, grads, m, v, float(lr[0]), int(step[0]), weight_decay, clip_value)

    def hard_update_every(self, target, source, step, interval):
        if int(step[0]) % interval == 0:
            target.copy_(source)

    # ---- one DDPG iteration on row blocks: the CPU double works from the row-major parameters (the packed copy is what
    # the HIP kernels read; a stale copy would go unnoticed here, so the double keeps its OWN snapshot, refreshed by
    # ddpg_rows_pack exactly as the packed copy is)
    def ddpg_rows_supported(self, D, A, H1, H2, c1, c2, rows=None):
        return all(v % 4 == 0 for v in (H1, H2, c1, c2)) and A <= 32

    def ddpg_rows_packed_floats(self, D, A, H1, H2, c1, c2):
        return 64

    def ddpg_rows_args(self, dims, nets, packed, io, gamma_n):
        import types
        a = types.SimpleNamespace(dims=dims, nets=nets, packed=packed, io=io, gamma_n=float(gamma_n), snap={})
        return a

    def ddpg_rows_pack(self, args, critic_only=False):
        for name in (('critic',) if critic_only else ('actor', 'critic', 'target_actor', 'target_critic')):
            args.snap[name] = {k: v.clone() for k, v in args.nets[name].items()}

    def ddpg_rows_update(self, args, group, theta, grads, exp_avg, exp_avg_sq, lr, step, weight_decay, clip_value,
                         target=None, tau=0.0, interval=0, wgrad=False, stats=None, stats_host=None):
        """Adam + the group's target update + BOTH snapshots (the kernel keeps the packed copies of the group and of its
        target current; nothing else is refreshed -- a schedule that relied on more would fail the goldens here).
        wgrad: the group's weight gradients first, from the buffers the chain launches wrote, into `grads` (laid out like
        theta: W1, b1, W2, b2, W3, b3)"""
        if wgrad:
            io = args.io
            D, A, H1, H2, c1, c2 = args.dims
            x = io['x']
            if group == 'critic':
                pairs = [(io['dxcat'][:, :c1], x), (io['dz2'], io['xcat']), (io['dz3'].view(-1, 1), io['h2c'])]
            else:
                pairs = [(io['dz1a'], x), (io['dz2a'], io['h1a']), (io['dz3a'], io['h2a'])]
            o = 0
            for dz, xin in pairs:
                M, N = dz.shape[1], xin.shape[1]
                grads[o:o + M * N].copy_((dz.t() @ xin).reshape(-1))
                o += M * N
                grads[o:o + M].copy_(dz.sum(0))
                o += M
            assert o == grads.numel()
        if stats is not None:       # (independent of the step: read from the chain launches' buffers)
            io = args.io
            self.ddpg_stats(io['q'], io['y'], io['rewards'], io['actions'], io['q_actor'], stats)
            if stats_host is not None:
                stats_host.view(2, 8)[int(step[0]) & 1, :7].copy_(stats[:7])
        self.adam_step_dev(theta, grads, exp_avg, exp_avg_sq, lr, step, weight_decay, clip_value)
        if target is not None:
            if interval > 0:
                self.hard_update_every(target, theta, step, interval)
            else:
                self.soft_update(target, theta, tau)
        for name in ((group, 'target_' + group) if target is not None else (group,)):
            args.snap[name] = {k: v.clone() for k, v in args.nets[name].items()}

    @staticmethod
    def _rows_actor_fwd(n, x):
        h1 = torch.relu(x @ n['W1'].t() + n['b1'])
        h2 = torch.relu(h1 @ n['W2'].t() + n['b2'])
        return h1, h2, torch.tanh(h2 @ n['W3'].t() + n['b3'])

    @staticmethod
    def _rows_critic_fwd(n, x, a):
        xcat = torch.cat([torch.relu(x @ n['W1'].t() + n['b1']), a], 1)
        h2 = torch.relu(xcat @ n['W2'].t() + n['b2'])
        return xcat, h2, (h2 @ n['W3'].t() + n['b3']).view(-1)

    def ddpg_rows_critic(self, args):
        io, S = args.io, args.snap
        D, A, H1, H2, c1, c2 = args.dims
        x, xn = io['x'], io['x_next']
        B = x.shape[0]
        _, _, a_next = self._rows_actor_fwd(S['target_actor'], xn)
        _, _, q_next = self._rows_critic_fwd(S['target_critic'], xn, a_next)
        xcat, h2c, q = self._rows_critic_fwd(S['critic'], x, io['actions'])
        y = io['rewards'].view(-1) + (args.gamma_n * q_next) * (1.0 - io['dones'].view(-1))
        dz3 = 2.0 * (q - y) / B
        io['xcat'].copy_(xcat); io['h2c'].copy_(h2c); io['q'].copy_(q); io['q_next'].copy_(q_next)
        io['y'].copy_(y); io['dz3'].copy_(dz3)
        if io.get('step') is not None:
            io['step'] += 1
        W2, W3 = S['critic']['W2'], S['critic']['W3']
        dz2 = (dz3.view(B, 1) * W3.view(1, c2)) * (h2c > 0)
        io['dz2'].copy_(dz2)
        io['dxcat'][:, :c1].copy_((dz2 @ W2[:, :c1]) * (xcat[:, :c1] > 0))
        h1a, h2a, act = self._rows_actor_fwd(S['actor'], x)
        io['h1a'].copy_(h1a); io['h2a'].copy_(h2a); io['act'].copy_(act)

    def ddpg_rows_actor(self, args):
        io, S = args.io, args.snap
        D, A, H1, H2, c1, c2 = args.dims
        x = io['x']
        B = x.shape[0]
        xcat, h2c, q = self._rows_critic_fwd(S['critic'], x, io['act'])
        io['q_actor'].copy_(q)
        W2, W3 = S['critic']['W2'], S['critic']['W3']
        dz2 = (torch.full((B, 1), -1.0 / B) * W3.view(1, c2)) * (h2c > 0)
        da = dz2 @ W2[:, c1:]
        dz3a = da * (1.0 - io['act'] * io['act'])
        io['dz3a'].copy_(dz3a)
        dz2a = (dz3a @ S['actor']['W3']) * (io['h2a'] > 0)
        io['dz2a'].copy_(dz2a)
        io['dz1a'].copy_((dz2a @ S['actor']['W2']) * (io['h1a'] > 0))

    def ddpg_stats(self, q, y, rewards, actions, q_actor, stats):
        stats[:6].copy_(_f([float(-q_actor.double().mean()), float(((q - y).double() ** 2).mean()),
                        float(actions.norm(2, 1).double().mean()), float(rewards.double().mean()),
                        float(y.double().mean()), float(q.double().mean())]))
        stats[6] = float('nan') if bool(torch.isnan(actions).any()) else actions.abs().max()
