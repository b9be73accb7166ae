// DiagGauss surrogate losses (forward + analytic backward), value loss, clip_grad_norm_ + Adam.
// Reference: surreal/model/ppo_net.py:29-72 (DiagGauss), surreal/learner/ppo.py:194-353
// (_clip_loss/_adapt_loss/_value_loss and their *_update methods), torch.optim.Adam
// (single-tensor path) and torch.nn.utils.clip_grad_norm_.
//
// The reference builds these out of ~40 ATen elementwise ops plus autograd and reads ~6 scalars
// back to the host per epoch (.item()).  Here each loss is ONE row-parallel kernel (one lane per
// row, wave-shuffle reductions into per-block partials) and one finalize kernel that turns the
// batch means into the gradient scale, the statistics and the device-side KL early-exit flag,
// so an epoch needs no host round trip and the whole learn() can be captured in a hipGraph.
#include "smx_common.h"
#include "smx_moments.inc.h"

namespace {
#include "smx_ppo_loss.inc.h"
#include "smx_epoch_pack.inc.h"
#include "smx_adam.inc.h"

__global__ __launch_bounds__(256) void policy_loss_kernel(
    int mode, const float* __restrict__ g_mean, const float* __restrict__ log_var,
    const float* __restrict__ g_actions, int ld_act, const float* __restrict__ g_behave, int ld_beh,
    const float* __restrict__ g_ref, int ld_ref, const float* __restrict__ adv, long rows, int A,
    const smx_ppo_ctrl_t* __restrict__ ctrl, float* __restrict__ g_surr, float* __restrict__ g_kl,
    float* __restrict__ partials) {
    if (ctrl->stop_flag) return;
    extern __shared__ float sm[];
    policy_loss_body(blockIdx.x, sm, mode, g_mean + (size_t)blockIdx.x * LOSS_ROWS_PER_BLOCK * A, A, log_var, g_actions, ld_act, g_behave, ld_beh, g_ref,
                     ld_ref, adv, rows, A, ctrl, g_surr, g_kl, partials);
}

// blk / nblocks: this workgroup's share of the elementwise part (blk 0 also writes the scalars)
__device__ __forceinline__ void policy_finalize_body(
    const int blk, const int nblocks,
    int mode, const float* __restrict__ partials, int nblk, const float* __restrict__ g_surr,
    const float* __restrict__ g_kl, const float* __restrict__ log_var, long rows, long n_total,
    int A, smx_ppo_ctrl_t* __restrict__ ctrl, int check_stop, int will_update,
    float* __restrict__ dz3, float* __restrict__ dz3_t, long ld_t, float* __restrict__ dlogvar,
    float* __restrict__ dlogvar_sumsq, float* __restrict__ stats) {
    __shared__ float S[8 + 2 * MAX_A];
    __shared__ float buf[FIN_CH * (8 + 2 * MAX_A)];
    reduce_row_partials(partials, nblk, 8 + 2 * A, S, buf);
    const float n = (float)n_total;
    float c_kl, loss;
    loss_and_kl_coef(mode, S, n, ctrl, loss, c_kl);
    const float inv_n = 1.0f / n;
    const long total = rows * A;
    for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)nblocks * 256)
    {
        const float v = (g_surr[i] + c_kl * g_kl[i]) * inv_n;
        dz3[i] = v;
        if (dz3_t) {
            const long r = i / A;
            dz3_t[(i - r * A) * ld_t + r] = v;
        }
    }
    if (blk == 0) {
        for (int a = threadIdx.x; a < A; a += 256)
            dlogvar[a] = (S[8 + a] + c_kl * S[8 + A + a]) * inv_n;
        if (threadIdx.x == 0)
            write_policy_scalars(S, n, loss, c_kl, log_var, A, ctrl, check_stop, will_update,
                                 dlogvar_sumsq, stats);
    }
}

// ---------------------------------------------------------------------------
// value loss: 256 rows per block; per-block mergeable moments of d = ret - V and of ret
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void policy_finalize_kernel(
    int mode, const float* __restrict__ partials, int nblk, const float* __restrict__ g_surr,
    const float* __restrict__ g_kl, const float* __restrict__ log_var, long rows, long n_total,
    int A, smx_ppo_ctrl_t* __restrict__ ctrl, int check_stop, int will_update,
    float* __restrict__ dz3, float* __restrict__ dz3_t, long ld_t, float* __restrict__ dlogvar,
    float* __restrict__ dlogvar_sumsq, float* __restrict__ stats) {
    // every argument is fetched together with the pointer to the stop flag (one kernarg round trip
    // in front of the flag's instead of one behind it as well)
    asm volatile("" :: "s"(mode), "s"(partials), "s"(nblk), "s"(g_surr), "s"(g_kl), "s"(log_var), "s"(rows),
                 "s"(n_total), "s"(A), "s"(ctrl), "s"(check_stop), "s"(will_update), "s"(dz3), "s"(dz3_t),
                 "s"(ld_t), "s"(dlogvar), "s"(dlogvar_sumsq), "s"(stats));
    if (ctrl->stop_flag) return;
    policy_finalize_body(blockIdx.x, gridDim.x, mode, partials, nblk, g_surr, g_kl, log_var, rows,
                         n_total, A, ctrl, check_stop, will_update, dz3, dz3_t, ld_t, dlogvar,
                         dlogvar_sumsq, stats);
}

__device__ __forceinline__ void value_loss_body(const int blk, const float* __restrict__ values,
                                                const float* __restrict__ returns, long rows,
                                                long n_total, float* __restrict__ dz3,
                                                float* __restrict__ partials,
                                                smx_ppo_ctrl_t* __restrict__ ctrl,
                                                int will_update) {
    __shared__ float red[16];
    const long r = (long)blk * 256 + threadIdx.x;
    const bool ok = r < rows;
    const float v = ok ? values[r] : 0.f, g = ok ? returns[r] : 0.f;
    const float d = g - v;                                   // returns - values (ppo.py:325)
    const float e = v - g;                                   // values - returns (ppo.py:326)
    if (ok) dz3[r] = (2.0f * e) / (float)n_total;
    long nb = rows - (long)blk * 256;
    if (nb > 256) nb = 256;
    const float cnt = (float)nb;
    const float md = smx_block_sum(ok ? d : 0.f, red) / cnt;
    const float mg = smx_block_sum(ok ? g : 0.f, red) / cnt;
    const float m2d = smx_block_sum(ok ? (d - md) * (d - md) : 0.f, red);
    const float m2g = smx_block_sum(ok ? (g - mg) * (g - mg) : 0.f, red);
    const float sq = smx_block_sum(ok ? e * e : 0.f, red);
    if (threadIdx.x == 0) {
        float* P = partials + (size_t)blk * 8;
        P[0] = cnt; P[1] = md; P[2] = m2d; P[3] = mg; P[4] = m2g; P[5] = sq; P[6] = 0.f; P[7] = 0.f;
        if (blk == 0 && will_update) ctrl->adam_step_critic += 1;
    }
}

__global__ __launch_bounds__(256) void value_loss_kernel(const float* __restrict__ values,
                                                         const float* __restrict__ returns,
                                                         long rows, long n_total,
                                                         float* __restrict__ dz3,
                                                         float* __restrict__ partials,
                                                         smx_ppo_ctrl_t* __restrict__ ctrl,
                                                         int will_update) {
    value_loss_body(blockIdx.x, values, returns, rows, n_total, dz3, partials, ctrl, will_update);
}

// The two row-parallel loss kernels of a lock-step epoch in one launch: workgroups [0, nblk_p)
// run the policy loss, workgroups [nblk_p, ...) the value loss (they share nothing).  The policy
// finalize stays its own multi-workgroup launch: folding it into the last workgroup to finish
// (device-scope ticket) was measured at 47 us per launch against 23 us for the three separate
// kernels -- one workgroup walking all rows x A gradient elements is far slower than a launch.
__global__ __launch_bounds__(256) void ppo_losses_kernel(smx_ppo_losses_t a,
                                                         smx_ppo_ctrl_t* __restrict__ ctrl,
                                                         int nblk_p, long n_total, int scaled,
                                                         float* __restrict__ g_surr_t,
                                                         float* __restrict__ g_kl_t) {
    extern __shared__ float sm[];
    asm volatile("" :: "s"(a.mode), "s"(a.A), "s"(a.mean), "s"(a.log_var), "s"(a.actions), "s"(a.behave),
                 "s"(a.ref), "s"(a.adv), "s"(a.ld_act), "s"(a.ld_beh), "s"(a.ld_ref), "s"(a.rows), "s"(a.g_surr),
                 "s"(a.g_kl), "s"(a.row_partials), "s"(a.ld_t), "s"(a.values), "s"(a.returns), "s"(a.v_dz3),
                 "s"(a.v_partials), "s"(a.v_will_update), "s"(ctrl), "s"(nblk_p), "s"(n_total), "s"(scaled),
                 "s"(g_surr_t), "s"(g_kl_t));             // one kernarg round trip for all of them
    if ((int)blockIdx.x >= nblk_p) {
        value_loss_body(blockIdx.x - nblk_p, a.values, a.returns, (long)a.rows, n_total, a.v_dz3,
                        a.v_partials, ctrl, a.v_will_update);
        return;
    }
    if (ctrl->stop_flag) return;
    policy_loss_body(blockIdx.x, sm, a.mode, a.mean + (size_t)blockIdx.x * LOSS_ROWS_PER_BLOCK * a.A, a.A, a.log_var, a.actions, a.ld_act, a.behave, a.ld_beh,
                     a.ref, a.ld_ref, a.adv, (long)a.rows, a.A, ctrl, a.g_surr, a.g_kl, a.row_partials,
                     1.0f / (float)n_total, scaled != 0, g_surr_t, g_kl_t, (long)a.ld_t);
}

// Data-parallel lock-step epoch, after the all-reduce of [surrogate share | critic gradient | KL
// share | loss partial rows]: the backward pass ran on the two right-hand sides g_surr / n and
// g_kl / n (it is linear in dz3), so the gradient of the loss is  G_surr + c_kl * G_kl  with c_kl
// from the GLOBAL mean KL -- formed here, together with log_var's gradient, the epoch statistics,
// the KL early exit and the sum-of-squares partials clip_grad_norm_ needs for both groups.
// Workgroups [0, nb_a): the actor group's elements; [nb_a, nb_a + nb_c): the critic's.
__global__ __launch_bounds__(256) void epoch_combine_kernel(smx_ppo_combine_t a,
                                                            smx_ppo_ctrl_t* __restrict__ ctrl,
                                                            int nb_a, int nb_c) {
    __shared__ float red[16];
    if ((int)blockIdx.x >= nb_a) {                       // critic: sum of squares only
        const int blk = blockIdx.x - nb_a;
        const long per = (a.n_c + nb_c - 1) / nb_c;
        const long lo = (long)blk * per;
        const long hi = min(lo + per, (long)a.n_c);
        float q = 0.f;
        for (long i = lo + threadIdx.x; i < hi; i += 256) q += a.grads_c[i] * a.grads_c[i];
        const float t = smx_block_sum(q, red);
        if (threadIdx.x == 0) a.sumsq_c[blk] = t;
        return;
    }
    if (ctrl->stop_flag) return;
    __shared__ float S[8 + 2 * MAX_A];
    __shared__ float buf[FIN_CH * (8 + 2 * MAX_A)];
    const int A = a.A;
    reduce_row_partials(a.row_partials, a.nblk, 8 + 2 * A, S, buf);
    const float n = (float)a.n_total;
    float c_kl, loss;
    loss_and_kl_coef(a.mode, S, n, ctrl, loss, c_kl);
    const float inv_n = 1.0f / n;
    const int blk = blockIdx.x;
    const long per = (a.n_a + nb_a - 1) / nb_a;
    const long lo = (long)blk * per;
    // the early exit taken by THIS epoch: workgroup 0 raises the flag below while the others may or
    // may not have read it yet, so all of them take the decision themselves -- no gradient is formed
    // (the Adam step is skipped anyway), whichever way the race goes
    const bool stop_now = a.check_stop && (double)(S[2] / n) > 4.0 * (double)ctrl->kl_target;
    const long hi = stop_now ? lo : min(lo + per, (long)a.n_a);
    float q = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        float g = a.grads_a[i];
        if (i < a.n_mlp) {
            if (a.grads_kl) { g = g + c_kl * a.grads_kl[i]; a.grads_a[i] = g; }
        } else if (i < a.n_mlp + A) {
            const int k = (int)(i - a.n_mlp);
            g = (S[8 + k] + c_kl * S[8 + A + k]) * inv_n;
            a.grads_a[i] = g;
        }
        q += g * g;
    }
    const float t = smx_block_sum(q, red);
    if (threadIdx.x == 0) {
        a.sumsq_a[blk] = t;
        if (blk == 0)
            write_policy_scalars(S, n, loss, c_kl, a.log_var, A, ctrl, a.check_stop, a.will_update,
                                 nullptr, a.stats);
    }
}

// one wave per epoch (smx_moments.inc.h)
__global__ __launch_bounds__(64) void value_finalize_kernel(const float* __restrict__ partials,
                                                            int count, int nblk,
                                                            float* __restrict__ stats,
                                                            int stats_stride) {
    const int e = blockIdx.x;
    if (e >= count) return;
    value_finalize_wave(partials + (size_t)e * nblk * 8, nblk, stats + (size_t)e * stats_stride, threadIdx.x);
}

// ---------------------------------------------------------------------------
// clip_grad_norm_ + Adam
// ---------------------------------------------------------------------------
struct AdamGroup {
    float* theta;
    const float* grads;
    float* m;
    float* v;
    long n;
    const float* partials;
    int npart, which, honour_stop, blocks;
    float* grad_norm_out;
    // optional: the fused epoch kernels' packed copy of the group's MLP, kept current by this step
    float* packed;
    long oW1, oW2, oW3;        // offsets of the three weight matrices inside theta
    int D, H1, H2, OUT;
};

struct AdamGroups {
    AdamGroup g[2];
    int n;
};

// one launch steps up to two optimiser groups (the actor's and the critic's of a lock-step epoch)
__global__ __launch_bounds__(256) void clip_adam_kernel(AdamGroups P,
                                                        const smx_ppo_ctrl_t* __restrict__ ctrl) {
    const int gi = (P.n > 1 && (int)blockIdx.x >= P.g[0].blocks) ? 1 : 0;
    // the group descriptor and the control block in one batch of scalar loads each (fetched field
    // by field at their first use they form a chain of six dependent round trips, ~1.5 us)
    const AdamGroup G = P.g[gi];
    asm volatile("" :: "s"(G.theta), "s"(G.grads), "s"(G.m), "s"(G.v), "s"(G.n), "s"(G.partials), "s"(G.npart),
                 "s"(G.which), "s"(G.honour_stop), "s"(G.blocks), "s"(G.grad_norm_out));
    const smx_ppo_ctrl_t C = *ctrl;
    asm volatile("" :: "s"(C.lr_actor), "s"(C.lr_critic), "s"(C.actor_max_norm), "s"(C.critic_max_norm),
                 "s"(C.actor_weight_decay), "s"(C.critic_weight_decay), "s"(C.adam_step_actor),
                 "s"(C.adam_step_critic), "s"(C.stop_flag));
    const smx_ppo_ctrl_t* ctrl_v = &C;
    const int blk = blockIdx.x - (gi ? P.g[0].blocks : 0);
    if (G.honour_stop && C.stop_flag) return;
    // reserved[0]: raised by a peer exchange of this learn() that timed out (smx_xchg.hip); reserved[1]: by the fused
    // forward + backward epoch launch whose in-launch wait timed out -- the gradients may then be anything: no step
    // for either group; the learner raises when it reads the words back
    if ((C.reserved[0] | C.reserved[1]) != 0) return;
    __shared__ float red[16];
    float* __restrict__ theta = G.theta;
    const float* __restrict__ grads = G.grads;
    float* __restrict__ m = G.m;
    float* __restrict__ v = G.v;
    // this thread's first two elements are requested in front of the norm reduction (independent of
    // it): one memory round trip for the step instead of two back to back
    const long stride = (long)G.blocks * 256;
    const long i0 = (long)blk * 256 + threadIdx.x, i1 = i0 + stride;
    const long c0 = i0 < G.n ? i0 : G.n - 1, c1 = i1 < G.n ? i1 : G.n - 1;
    const float g0 = grads[c0], p0 = theta[c0], m0 = m[c0], v0 = v[c0];
    const float g1 = grads[c1], p1 = theta[c1], m1 = m[c1], v1 = v[c1];
    float t = 0.f;
    for (int k = threadIdx.x; k < G.npart; k += 256) t += G.partials[k];
    const float total = smx_block_sum(t, red);
    const float norm = sqrtf(total);
    const int which = G.which;
    if (blk == 0 && threadIdx.x == 0 && G.grad_norm_out) *G.grad_norm_out = norm;
    const AdamCoef K = adam_coef(*ctrl_v, which, norm);
    AdamPack Q;
    Q.packed = G.packed; Q.oW1 = G.oW1; Q.oW2 = G.oW2; Q.oW3 = G.oW3; Q.D = G.D; Q.H1 = G.H1; Q.H2 = G.H2; Q.OUT = G.OUT;
    auto step_one = [&](long i, float g, float p, float mi, float vi) { adam_step_one(K, Q, theta, m, v, i, g, p, mi, vi); };
    if (i0 < G.n) step_one(i0, g0, p0, m0, v0);
    if (i1 < G.n) step_one(i1, g1, p1, m1, v1);
    for (long i = i1 + stride; i < G.n; i += stride) step_one(i, grads[i], theta[i], m[i], v[i]);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n,
                                                    float* __restrict__ partials) {
    __shared__ float red[16];
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per;
    long hi = lo + per;
    if (hi > n) hi = n;
    float s = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) s += x[i] * x[i];
    const float t = smx_block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void final_stats_kernel(const float* __restrict__ log_var, int A,
                                                          const float* __restrict__ rsum,
                                                          const float* __restrict__ rsumsq,
                                                          const float* __restrict__ count, int D,
                                                          float* __restrict__ out) {
    __shared__ double red[16][4];
    final_stats_block(log_var, A, rsum, rsumsq, count, D, out, red);
}

// acting head of PPOAgent.act (ppo_agent.py:106-154, ppo_net.py:74-91): pd = [mean, exp(log_var) *
// noise_r], action = clip(eps * std + mean, -1, 1) (training) or clip(mean) (eps == NULL)
__global__ __launch_bounds__(256) void diaggauss_sample_kernel(const float* __restrict__ mean, long ld_mean,
                                                               const float* __restrict__ log_var,
                                                               const float* __restrict__ noise,
                                                               const float* __restrict__ eps, long ld_eps,
                                                               long total, int A,
                                                               float* __restrict__ actions, long ld_act,
                                                               float* __restrict__ pd, long ld_pd) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / A;
        const int a = (int)(i - r * A);
        const float mu = mean[r * ld_mean + a];
        float sd = expf(log_var[a]);                        // exp(log_var) * ones_like(mean)
        if (noise) sd = sd * noise[r];                      // action_pd[:, A:] *= exp(noise)
        float act = eps ? eps[r * ld_eps + a] * sd + mu : mu;
        if (act == act) act = fminf(fmaxf(act, -1.0f), 1.0f);
        actions[r * ld_act + a] = act;
        if (pd) {
            pd[r * ld_pd + a] = mu;
            pd[r * ld_pd + A + a] = sd;
        }
    }
}

}  // namespace

extern "C" int smx_ppo_final_stats_f32(const float* log_var, int32_t A, const float* running_sum,
                                       const float* running_sumsq, const float* count, int32_t D,
                                       float* out4, smx_stream_t stream) {
    SMX_REQUIRE(log_var && out4, SMX_E_NULL);
    SMX_REQUIRE(A > 0 && (running_sum == nullptr || (running_sumsq && count && D > 0)), SMX_E_SHAPE);
    hipLaunchKernelGGL(final_stats_kernel, dim3(1), dim3(256), 0, smx_s(stream), log_var, A,
                       running_sum, running_sumsq, count, D, out4);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_diaggauss_sample_f32(const float* mean, int64_t ld_mean, const float* log_var,
                                        const float* noise_scale, const float* eps, int64_t ld_eps,
                                        int64_t rows, int32_t A, float* actions, int64_t ld_act,
                                        float* pd, int64_t ld_pd, smx_stream_t stream) {
    SMX_REQUIRE(mean && log_var && actions, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && A > 0 && ld_mean >= A && ld_act >= A && (!eps || ld_eps >= A) &&
                    (!pd || ld_pd >= 2 * A), SMX_E_SHAPE);
    const long total = (long)rows * A;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(diaggauss_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), mean,
                       (long)ld_mean, log_var, noise_scale, eps, (long)ld_eps, total, A, actions,
                       (long)ld_act, pd, (long)ld_pd);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int32_t smx_ppo_loss_blocks(int64_t rows) {
    return (int32_t)((rows + LOSS_ROWS_PER_BLOCK - 1) / LOSS_ROWS_PER_BLOCK);
}
extern "C" int32_t smx_ppo_loss_partial_stride(int32_t A) { return 8 + 2 * A; }

extern "C" int smx_ppo_policy_loss_f32(int32_t mode, const float* mean, const float* log_var,
                                       const float* actions, int32_t ld_act, const float* behave,
                                       int32_t ld_beh, const float* ref, int32_t ld_ref,
                                       const float* adv, int64_t rows, int32_t A,
                                       const smx_ppo_ctrl_t* ctrl, float* g_surr, float* g_kl,
                                       float* row_partials, smx_stream_t stream) {
    SMX_REQUIRE(mean && log_var && actions && behave && ref && adv && ctrl && g_surr && g_kl &&
                    row_partials, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && A > 0 && ld_act >= A && ld_beh >= 2 * A && ld_ref >= 2 * A, SMX_E_SHAPE);
    SMX_REQUIRE(A <= MAX_A && (mode == SMX_PPO_CLIP || mode == SMX_PPO_ADAPT), SMX_E_UNSUPPORTED);
    const size_t lds = (size_t)loss_scratch_floats(A) * sizeof(float);
    hipLaunchKernelGGL(policy_loss_kernel, dim3(smx_ppo_loss_blocks(rows)), dim3(256), lds,
                       smx_s(stream), mode, mean, log_var, actions, ld_act, behave, ld_beh, ref,
                       ld_ref, adv, (long)rows, A, ctrl, g_surr, g_kl, row_partials);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// Many partial rows (the stems' policies: 7936 at 126 976 rows) folded to `nout` in front of the finalize, whose every
// workgroup otherwise walks all of them (143 us per launch measured there, 124 staged chunks): workgroup j adds the rows
// [j R, (j + 1) R), R = ceil(nblk / nout), column c by Q = 256 / stride chains (rows q, q + Q, ...) that meet in the fixed
// order ((s0 + s1) + s2) + ... -- the same bits on every run.
__global__ __launch_bounds__(256) void partials_fold_kernel(const float* __restrict__ src, int nblk, int stride,
                                                            float* __restrict__ dst, int nout, const smx_ppo_ctrl_t* ctrl) {
    if (ctrl && ctrl->stop_flag) return;
    __shared__ float buf[256];
    const int R = (nblk + nout - 1) / nout;
    const int r0 = blockIdx.x * R, r1 = min(nblk, r0 + R);
    const int Q = 256 / stride;
    const int q = (int)threadIdx.x / stride, c = (int)threadIdx.x - q * stride;
    float t0 = 0.f, t1 = 0.f;
    if (q < Q) {
        int r = r0 + q;
        for (; r + Q < r1; r += 2 * Q) {
            t0 += src[(size_t)r * stride + c];
            t1 += src[(size_t)(r + Q) * stride + c];
        }
        if (r < r1) t0 += src[(size_t)r * stride + c];
    }
    buf[threadIdx.x] = t0 + t1;
    __syncthreads();
    if ((int)threadIdx.x < stride) {
        float sum = buf[threadIdx.x];
        for (int k = 1; k < Q; ++k) sum += buf[k * stride + threadIdx.x];
        dst[(size_t)blockIdx.x * stride + threadIdx.x] = sum;
    }
}

extern "C" int smx_ppo_partials_fold_f32(const float* row_partials, int32_t nblk, int32_t stride, float* out, int32_t nout,
                                         const smx_ppo_ctrl_t* ctrl, smx_stream_t stream) {
    SMX_REQUIRE(row_partials && out, SMX_E_NULL);
    SMX_REQUIRE(nblk > 0 && nout > 0 && nout <= nblk && stride > 0 && stride <= 128, SMX_E_SHAPE);
    hipLaunchKernelGGL(partials_fold_kernel, dim3((unsigned)nout), dim3(256), 0, smx_s(stream), row_partials, nblk, stride, out,
                       nout, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ppo_loss_finalize_f32(int32_t mode, const float* row_partials, int32_t nblk,
                                         const float* g_surr, const float* g_kl,
                                         const float* log_var, int64_t rows, int64_t n_total,
                                         int32_t A, smx_ppo_ctrl_t* ctrl, int32_t check_stop,
                                         int32_t will_update, float* dz3, float* dz3_t,
                                         int64_t ld_t, float* dlogvar, float* dlogvar_sumsq, float* stats,
                                         smx_stream_t stream) {
    SMX_REQUIRE(row_partials && g_surr && g_kl && log_var && ctrl && dz3 && dlogvar && stats,
                SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && n_total >= rows && A > 0 && nblk > 0, SMX_E_SHAPE);
    SMX_REQUIRE(A <= MAX_A, SMX_E_UNSUPPORTED);
    long blocks = (rows * A + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(policy_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream),
                       mode, row_partials, nblk, g_surr, g_kl, log_var, (long)rows, (long)n_total, A,
                       ctrl, check_stop, will_update, dz3, dz3_t, (long)(ld_t ? ld_t : rows), dlogvar,
                       dlogvar_sumsq, stats);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int32_t smx_value_loss_blocks(int64_t rows) { return (int32_t)((rows + 255) / 256); }

extern "C" int smx_value_loss_f32(const float* values, const float* returns, int64_t rows,
                                  int64_t n_total, float* dz3, float* partials,
                                  smx_ppo_ctrl_t* ctrl, int32_t will_update, smx_stream_t stream) {
    SMX_REQUIRE(values && returns && dz3 && partials && ctrl, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && n_total >= rows, SMX_E_SHAPE);
    hipLaunchKernelGGL(value_loss_kernel, dim3(smx_value_loss_blocks(rows)), dim3(256), 0,
                       smx_s(stream), values, returns, (long)rows, (long)n_total, dz3, partials, ctrl,
                       will_update);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ppo_epoch_losses_f32(const smx_ppo_losses_t* args, smx_ppo_ctrl_t* ctrl,
                                        smx_stream_t stream) {
    SMX_REQUIRE(args && ctrl, SMX_E_NULL);
    const smx_ppo_losses_t& a = *args;
    SMX_REQUIRE(a.mean && a.log_var && a.actions && a.behave && a.ref && a.adv && a.g_surr && a.g_kl &&
                    a.row_partials && a.dz3 && a.dlogvar && a.stats, SMX_E_NULL);
    SMX_REQUIRE(a.rows > 0 && a.A > 0 && a.ld_act >= a.A && a.ld_beh >= 2 * a.A && a.ld_ref >= 2 * a.A,
                SMX_E_SHAPE);
    SMX_REQUIRE(a.A <= MAX_A && (a.mode == SMX_PPO_CLIP || a.mode == SMX_PPO_ADAPT), SMX_E_UNSUPPORTED);
    int nblk_v = 0;
    if (a.values) {
        SMX_REQUIRE(a.returns && a.v_dz3 && a.v_partials, SMX_E_NULL);
        nblk_v = smx_value_loss_blocks(a.rows);
    }
    const int nblk_p = smx_ppo_loss_blocks(a.rows);
    const size_t lds = (size_t)loss_scratch_floats(a.A) * sizeof(float);
    hipLaunchKernelGGL(ppo_losses_kernel, dim3(nblk_p + nblk_v), dim3(256), lds, smx_s(stream), a, ctrl,
                       nblk_p, (long)a.rows, 0, (float*)nullptr, (float*)nullptr);
    SMX_LAUNCH_CHECK();
    return smx_ppo_loss_finalize_f32(a.mode, a.row_partials, nblk_p, a.g_surr, a.g_kl, a.log_var, a.rows,
                                     a.rows, a.A, ctrl, a.check_stop, a.will_update, a.dz3, a.dz3_t,
                                     a.ld_t, a.dlogvar, a.dlogvar_sumsq, a.stats, stream);
}

extern "C" int smx_ppo_epoch_losses_dp_f32(const smx_ppo_losses_t* args, int64_t n_total,
                                           float* g_surr_t, float* g_kl_t, smx_ppo_ctrl_t* ctrl,
                                           smx_stream_t stream) {
    SMX_REQUIRE(args && ctrl, SMX_E_NULL);
    const smx_ppo_losses_t& a = *args;
    SMX_REQUIRE(a.mean && a.log_var && a.actions && a.behave && a.ref && a.adv && a.g_surr && a.g_kl &&
                    a.row_partials, SMX_E_NULL);
    SMX_REQUIRE(a.rows > 0 && n_total >= a.rows && a.A > 0 && a.ld_act >= a.A && a.ld_beh >= 2 * a.A &&
                    a.ld_ref >= 2 * a.A, SMX_E_SHAPE);
    SMX_REQUIRE((g_surr_t == nullptr && g_kl_t == nullptr) || (g_surr_t && a.ld_t >= a.rows), SMX_E_SHAPE);
    SMX_REQUIRE(a.A <= MAX_A && (a.mode == SMX_PPO_CLIP || a.mode == SMX_PPO_ADAPT), SMX_E_UNSUPPORTED);
    int nblk_v = 0;
    if (a.values) {
        SMX_REQUIRE(a.returns && a.v_dz3 && a.v_partials, SMX_E_NULL);
        nblk_v = smx_value_loss_blocks(a.rows);
    }
    const int nblk_p = smx_ppo_loss_blocks(a.rows);
    const size_t lds = (size_t)loss_scratch_floats(a.A) * sizeof(float);
    hipLaunchKernelGGL(ppo_losses_kernel, dim3(nblk_p + nblk_v), dim3(256), lds, smx_s(stream), a, ctrl,
                       nblk_p, (long)n_total, 1, g_surr_t, g_kl_t);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ppo_epoch_combine_f32(const smx_ppo_combine_t* args, smx_ppo_ctrl_t* ctrl,
                                         smx_stream_t stream) {
    SMX_REQUIRE(args && ctrl, SMX_E_NULL);
    const smx_ppo_combine_t& a = *args;
    SMX_REQUIRE(a.row_partials && a.log_var && a.stats && a.grads_a && a.sumsq_a, SMX_E_NULL);
    SMX_REQUIRE(a.A > 0 && a.nblk > 0 && a.n_total > 0 && a.n_mlp >= 0 && a.n_a >= a.n_mlp + a.A,
                SMX_E_SHAPE);
    SMX_REQUIRE(a.A <= MAX_A && (a.mode == SMX_PPO_CLIP || a.mode == SMX_PPO_ADAPT), SMX_E_UNSUPPORTED);
    SMX_REQUIRE(a.mode == SMX_PPO_CLIP || a.grads_kl, SMX_E_NULL);
    int nb_c = 0;
    if (a.grads_c) {
        SMX_REQUIRE(a.sumsq_c && a.n_c > 0, SMX_E_NULL);
        nb_c = smx_sumsq_blocks(a.n_c);
    }
    const int nb_a = smx_sumsq_blocks(a.n_a);
    hipLaunchKernelGGL(epoch_combine_kernel, dim3(nb_a + nb_c), dim3(256), 0, smx_s(stream), a, ctrl,
                       nb_a, nb_c);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_value_loss_finalize_f32(const float* partials, int32_t count, int32_t nblk,
                                           float* stats, int32_t stats_stride,
                                           smx_stream_t stream) {
    SMX_REQUIRE(partials && stats, SMX_E_NULL);
    SMX_REQUIRE(count > 0 && nblk > 0 && stats_stride >= 2, SMX_E_SHAPE);
    hipLaunchKernelGGL(value_finalize_kernel, dim3(count), dim3(64), 0, smx_s(stream),
                       partials, count, nblk, stats, stats_stride);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

static int fill_adam_group(AdamGroup& G, float* theta, const float* grads, float* exp_avg,
                           float* exp_avg_sq, int64_t n, const float* sumsq_partials,
                           int32_t npart, int32_t which, int32_t honour_stop, float* grad_norm_out,
                           const smx_mlp3_t* pack_net = nullptr, float* packed = nullptr) {
    G.packed = nullptr;
    G.oW1 = G.oW2 = G.oW3 = 0;
    G.D = G.H1 = G.H2 = G.OUT = 0;
    if (packed) {
        SMX_REQUIRE(pack_net && theta, SMX_E_NULL);
        const smx_mlp3_t& N = *pack_net;
        G.packed = packed;
        G.oW1 = N.W1 - theta; G.oW2 = N.W2 - theta; G.oW3 = N.W3 - theta;
        G.D = N.D; G.H1 = N.H1; G.H2 = N.H2; G.OUT = N.OUT;
        SMX_REQUIRE(G.oW1 >= 0 && G.oW3 + (long)N.OUT * N.H2 <= n, SMX_E_SHAPE);   // the MLP lies inside theta
    }
    SMX_REQUIRE(theta && grads && exp_avg && exp_avg_sq && sumsq_partials, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && npart > 0, SMX_E_SHAPE);
    long blocks = (n + 511) / 512;          // two elements per thread
    if (blocks > 1024) blocks = 1024;
    G.theta = theta; G.grads = grads; G.m = exp_avg; G.v = exp_avg_sq; G.n = (long)n;
    G.partials = sumsq_partials; G.npart = npart; G.which = which; G.honour_stop = honour_stop;
    G.blocks = (int)blocks; G.grad_norm_out = grad_norm_out;
    return SMX_OK;
}

extern "C" int smx_clip_adam_step_f32(float* theta, const float* grads, float* exp_avg,
                                      float* exp_avg_sq, int64_t n, const float* sumsq_partials,
                                      int32_t npart, const smx_ppo_ctrl_t* ctrl, int32_t which,
                                      int32_t honour_stop, float* grad_norm_out,
                                      smx_stream_t stream) {
    SMX_REQUIRE(ctrl, SMX_E_NULL);
    AdamGroups P;
    P.n = 1;
    const int rc = fill_adam_group(P.g[0], theta, grads, exp_avg, exp_avg_sq, n, sumsq_partials,
                                   npart, which, honour_stop, grad_norm_out);
    if (rc) return rc;
    P.g[1] = P.g[0];
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)P.g[0].blocks), dim3(256), 0,
                       smx_s(stream), P, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_clip_adam_step_group_f32(const smx_adam_group_t* group, int32_t which,
                                            const smx_ppo_ctrl_t* ctrl, smx_stream_t stream) {
    SMX_REQUIRE(group && ctrl, SMX_E_NULL);
    SMX_REQUIRE(which == 0 || which == 1, SMX_E_SHAPE);
    AdamGroups P;
    P.n = 1;
    const int rc = fill_adam_group(P.g[0], group->theta, group->grads, group->exp_avg, group->exp_avg_sq,
                                   group->n, group->sumsq_partials, group->npart, which, group->honour_stop,
                                   group->grad_norm_out, group->pack_net, group->packed);
    if (rc) return rc;
    P.g[1] = P.g[0];
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)P.g[0].blocks), dim3(256), 0, smx_s(stream), P, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_clip_adam_step_pair_f32(const smx_adam_group_t* actor,
                                           const smx_adam_group_t* critic,
                                           const smx_ppo_ctrl_t* ctrl, smx_stream_t stream) {
    SMX_REQUIRE(actor && critic && ctrl, SMX_E_NULL);
    AdamGroups P;
    P.n = 2;
    const smx_adam_group_t* src[2] = {actor, critic};
    for (int k = 0; k < 2; ++k) {
        const int rc = fill_adam_group(P.g[k], src[k]->theta, src[k]->grads, src[k]->exp_avg,
                                       src[k]->exp_avg_sq, src[k]->n, src[k]->sumsq_partials,
                                       src[k]->npart, k, src[k]->honour_stop,
                                       src[k]->grad_norm_out, src[k]->pack_net, src[k]->packed);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)(P.g[0].blocks + P.g[1].blocks)),
                       dim3(256), 0, smx_s(stream), P, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int32_t smx_sumsq_blocks(int64_t n) {
    long b = (n + 4095) / 4096;
    if (b > 256) b = 256;
    if (b < 1) b = 1;
    return (int32_t)b;
}

extern "C" int smx_sumsq_partials_f32(const float* x, int64_t n, float* partials,
                                      smx_stream_t stream) {
    SMX_REQUIRE(x && partials, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(sumsq_kernel, dim3(smx_sumsq_blocks(n)), dim3(256), 0, smx_s(stream), x,
                       (long)n, partials);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
