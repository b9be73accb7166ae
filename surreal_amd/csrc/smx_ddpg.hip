// DDPG update pieces that are not plain dense layers (surreal/learner/ddpg.py:244-352,403-428):
// Bellman target + MSE gradient, tanh backward, value-clipped Adam, soft/hard target update,
// the statistics the reference logs.  All HBM-bound elementwise / small-reduction kernels; the
// dense layers (actor MLP, critic with the action concatenated into layer 2) reuse the FP32-MFMA
// smx_linear_f32 / smx_mlp3_* entry points.
#include "smx_common.h"

namespace {

__global__ __launch_bounds__(256) void ddpg_critic_loss_kernel(
    const float* __restrict__ q, const float* __restrict__ q_next, const float* __restrict__ rewards,
    const float* __restrict__ dones, float gamma_n, long rows, float* __restrict__ y,
    float* __restrict__ dz3, int* __restrict__ step_counter) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r == 0 && step_counter) *step_counter += 1;      // this iteration's Adam step (both groups)
    if (r >= rows) return;
    // y = rewards + gamma^n * Q'(s', mu'(s')) * (1 - done)        (ddpg.py:279)
    const float t = (gamma_n * q_next[r]) * (1.0f - dones[r]);
    const float yy = rewards[r] + t;
    y[r] = yy;
    dz3[r] = (2.0f * (q[r] - yy)) / (float)rows;   // d MSELoss / dQ   (ddpg.py:307-308)
}

__global__ __launch_bounds__(256) void tanh_backward_kernel(const float* __restrict__ da,
                                                            const float* __restrict__ a, long n,
                                                            float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = da[i] * (1.0f - a[i] * a[i]);
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, long n, float v) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = v;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ theta,
                                                   const float* __restrict__ grads,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   float neg_step_size, float bc2_sqrt, float w1,
                                                   float b2f, float w2, float eps, float wd,
                                                   float clip_value) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = grads[i];
    if (clip_value > 0.f) g = fminf(fmaxf(g, -clip_value), clip_value);   // clip_grad_value_
    const float p = theta[i];
    if (wd != 0.f) g = g + wd * p;
    float mi = m[i], vi = v[i];
    mi = mi + w1 * (g - mi);
    vi = vi * b2f + w2 * (g * g);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    theta[i] = p + (neg_step_size * mi) / denom;
    m[i] = mi;
    v[i] = vi;
}

// the same step with the learning rate and the step count read from device memory, so that a
// captured hipGraph of the whole iteration can be replayed while they change
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ theta,
                                                       const float* __restrict__ grads,
                                                       float* __restrict__ m, float* __restrict__ v, long n,
                                                       const float* __restrict__ lr_ptr,
                                                       const int* __restrict__ step_ptr, float wd,
                                                       float clip_value) {
    __shared__ float coef[2];
    if (threadIdx.x == 0) {
        const double beta1 = 0.9, beta2 = 0.999;
        const double step = (double)*step_ptr;
        const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
        coef[0] = (float)(-((double)*lr_ptr / bc1));
        coef[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float neg_step_size = coef[0], bc2_sqrt = coef[1];
    const float w1 = (float)(1.0 - 0.9), b2f = (float)0.999, w2 = (float)(1.0 - 0.999), eps = 1e-8f;
    float g = grads[i];
    if (clip_value > 0.f) g = fminf(fmaxf(g, -clip_value), clip_value);   // clip_grad_value_
    const float p = theta[i];
    if (wd != 0.f) g = g + wd * p;
    float mi = m[i], vi = v[i];
    mi = mi + w1 * (g - mi);
    vi = vi * b2f + w2 * (g * g);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    theta[i] = p + (neg_step_size * mi) / denom;
    m[i] = mi;
    v[i] = vi;
}

// hard target update every `interval` iterations (ddpg.py:403-409), decided on the device
__global__ __launch_bounds__(256) void hard_update_dev_kernel(float* __restrict__ tgt,
                                                              const float* __restrict__ src, long n,
                                                              const int* __restrict__ step_ptr,
                                                              int interval) {
    if (*step_ptr % interval != 0) return;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) tgt[i] = src[i];
}

__global__ __launch_bounds__(256) void soft_update_kernel(float* __restrict__ tgt,
                                                          const float* __restrict__ src, float tau,
                                                          long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // torchx Module.soft_update: target = target * (1 - tau) + source * tau ; tau = 1 -> hard copy
    tgt[i] = (tau >= 1.0f) ? src[i] : (tgt[i] * (1.0f - tau) + src[i] * tau);
}

// stats[6] = {actor_loss, critic_loss, action_norm, rewards, Q_target, Q_policy}  (ddpg.py:335-342)
#include "smx_ddpg_stats.inc.h"
__global__ __launch_bounds__(1024) void ddpg_stats_kernel(
    const float* __restrict__ q, const float* __restrict__ y, const float* __restrict__ rewards,
    const float* __restrict__ actions, int ld_act, int A, const float* __restrict__ q_actor, long rows,
    float* __restrict__ stats) {
    ddpg_stats_block<1024>(q, y, rewards, actions, ld_act, A, q_actor, rows, stats);
}

inline unsigned nb(long n) { return (unsigned)((n + 255) / 256); }


// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension (DDPG's use_layernorm = True: L.LayerNorm(1) behind every hidden ReLU of
// ActorNetworkX / CriticNetworkX, surreal/model/model_builders/builders.py:42-48, 65-75; torchx's layer is taken as
// torch.nn.LayerNorm(F): biased variance, eps inside the square root, elementwise affine -- its source is not in the
// reference tree).  One wavefront per row, two passes over the row in registers.
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAXC = 16;      // columns per lane: F <= 1024

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, long ldx, long rows, int F,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float* __restrict__ y, long ldy,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + r * ldx;
    float v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int j = lane + 64 * c;
        v[c] = (j < F) ? xr[j] : 0.f;
        s += v[c];
    }
    const float m = smx_wave_sum(s) / (float)F;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int j = lane + 64 * c;
        const float d = (j < F) ? v[c] - m : 0.f;
        q += d * d;
    }
    const float rs = 1.0f / sqrtf(smx_wave_sum(q) / (float)F + eps);
    float* yr = y + r * ldy;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int j = lane + 64 * c;
        if (j < F) yr[j] = ((v[c] - m) * rs) * gamma[j] + beta[j];
    }
    if (lane == 0) {
        if (mean) mean[r] = m;
        if (rstd) rstd[r] = rs;
    }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma, xhat = (x - mean) rstd; optionally times (x > 0): x is the
// output of the ReLU in front of the LayerNorm, so the product is the gradient at the ReLU's input.  Parameter gradients:
// every block leaves the column sums of dy xhat and dy over ITS 16 rows in part[blk][2][F]; layernorm_pgrad_kernel adds
// the blocks in order (deterministic).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                            long ldx, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            long rows, int F, int relu_mask, float* __restrict__ dx, long lddx,
                                                            float* __restrict__ part) {
    __shared__ float red[4][2][64 * LN_MAXC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float ag[LN_MAXC], ab[LN_MAXC];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) { ag[c] = 0.f; ab[c] = 0.f; }
    for (int k = 0; k < 4; ++k) {
        const long r = (long)blockIdx.x * 16 + 4 * k + wv;           // (wave-uniform)
        if (r >= rows) break;
        const float m = mean[r], rs = rstd[r];
        const float* xr = x + r * ldx;
        const float* dr = dy + r * lddy;
        float xv[LN_MAXC], g[LN_MAXC], xh[LN_MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int j = lane + 64 * c;
            const bool in = j < F;
            xv[c] = in ? xr[j] : 0.f;
            const float d = in ? dr[j] : 0.f;
            xh[c] = in ? (xv[c] - m) * rs : 0.f;
            g[c] = in ? d * gamma[j] : 0.f;
            s1 += g[c];
            s2 += g[c] * xh[c];
            ag[c] += d * xh[c];
            ab[c] += d;
        }
        const float m1 = smx_wave_sum(s1) / (float)F, m2 = smx_wave_sum(s2) / (float)F;
        float* o = dx + r * lddx;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int j = lane + 64 * c;
            if (j < F) {
                float v = rs * ((g[c] - m1) - xh[c] * m2);
                if (relu_mask) v = (xv[c] > 0.f) ? v : 0.f;
                o[j] = v;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        red[wv][0][lane + 64 * c] = ag[c];
        red[wv][1][lane + 64 * c] = ab[c];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < F; j += 256) {
        part[((size_t)blockIdx.x * 2 + 0) * F + j] = ((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j];
        part[((size_t)blockIdx.x * 2 + 1) * F + j] = ((red[0][1][j] + red[1][1][j]) + red[2][1][j]) + red[3][1][j];
    }
}

__global__ __launch_bounds__(256) void layernorm_pgrad_kernel(const float* __restrict__ part, int nblk, int F,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < F; j += gridDim.x * 256) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < nblk; ++k) {
            a += part[((size_t)k * 2 + 0) * F + j];
            b += part[((size_t)k * 2 + 1) * F + j];
        }
        dgamma[j] = a;
        dbeta[j] = b;
    }
}

}  // namespace

extern "C" int smx_ddpg_critic_loss_f32(const float* q, const float* q_next_target,
                                        const float* rewards, const float* dones, float gamma_n,
                                        int64_t rows, float* y, float* dz3, smx_stream_t stream) {
    SMX_REQUIRE(q && q_next_target && rewards && dones && y && dz3, SMX_E_NULL);
    SMX_REQUIRE(rows > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(ddpg_critic_loss_kernel, dim3(nb(rows)), dim3(256), 0, smx_s(stream), q,
                       q_next_target, rewards, dones, gamma_n, (long)rows, y, dz3, (int*)nullptr);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ddpg_critic_loss_step_f32(const float* q, const float* q_next_target,
                                             const float* rewards, const float* dones, float gamma_n,
                                             int64_t rows, float* y, float* dz3, int32_t* step_counter,
                                             smx_stream_t stream) {
    SMX_REQUIRE(q && q_next_target && rewards && dones && y && dz3 && step_counter, SMX_E_NULL);
    SMX_REQUIRE(rows > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(ddpg_critic_loss_kernel, dim3(nb(rows)), dim3(256), 0, smx_s(stream), q,
                       q_next_target, rewards, dones, gamma_n, (long)rows, y, dz3, step_counter);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_tanh_backward_f32(const float* da, const float* a, int64_t n, float* out,
                                     smx_stream_t stream) {
    SMX_REQUIRE(da && a && out, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(tanh_backward_kernel, dim3(nb(n)), dim3(256), 0, smx_s(stream), da, a, (long)n,
                       out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_fill_f32(float* x, int64_t n, float value, smx_stream_t stream) {
    SMX_REQUIRE(x, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(fill_kernel, dim3(nb(n)), dim3(256), 0, smx_s(stream), x, (long)n, value);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_adam_step_f32(float* theta, const float* grads, float* exp_avg, float* exp_avg_sq,
                                 int64_t n, double lr, int32_t step, double weight_decay,
                                 double clip_value, smx_stream_t stream) {
    SMX_REQUIRE(theta && grads && exp_avg && exp_avg_sq, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && step > 0, SMX_E_SHAPE);
    const double beta1 = 0.9, beta2 = 0.999;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(nb(n)), dim3(256), 0, smx_s(stream), theta, grads, exp_avg,
                       exp_avg_sq, (long)n, (float)(-(lr / bc1)), (float)sqrt(bc2), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), 1e-8f, (float)weight_decay, (float)clip_value);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_adam_step_dev_f32(float* theta, const float* grads, float* exp_avg,
                                     float* exp_avg_sq, int64_t n, const float* lr,
                                     const int32_t* step, double weight_decay, double clip_value,
                                     smx_stream_t stream) {
    SMX_REQUIRE(theta && grads && exp_avg && exp_avg_sq && lr && step, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(nb(n)), dim3(256), 0, smx_s(stream), theta, grads, exp_avg,
                       exp_avg_sq, (long)n, lr, step, (float)weight_decay, (float)clip_value);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_hard_update_every_f32(float* target, const float* source, int64_t n,
                                         const int32_t* step, int32_t interval, smx_stream_t stream) {
    SMX_REQUIRE(target && source && step, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && interval > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(hard_update_dev_kernel, dim3(nb(n)), dim3(256), 0, smx_s(stream), target, source,
                       (long)n, step, interval);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_soft_update_f32(float* target, const float* source, float tau, int64_t n,
                                   smx_stream_t stream) {
    SMX_REQUIRE(target && source, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && tau > 0.f, SMX_E_SHAPE);
    hipLaunchKernelGGL(soft_update_kernel, dim3(nb(n)), dim3(256), 0, smx_s(stream), target, source,
                       tau, (long)n);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ddpg_stats_f32(const float* q, const float* y, const float* rewards,
                                  const float* actions, int32_t ld_act, int32_t A,
                                  const float* q_actor, int64_t rows, float* stats,
                                  smx_stream_t stream) {
    SMX_REQUIRE(q && y && rewards && actions && q_actor && stats, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && A > 0 && ld_act >= A, SMX_E_SHAPE);
    hipLaunchKernelGGL(ddpg_stats_kernel, dim3(1), dim3(1024), 0, smx_s(stream), q, y, rewards, actions,
                       ld_act, A, q_actor, (long)rows, stats);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_layernorm_forward_f32(const float* x, int64_t ldx, int64_t rows, int32_t F, const float* gamma,
                                         const float* beta, float eps, float* y, int64_t ldy, float* mean, float* rstd,
                                         smx_stream_t stream) {
    SMX_REQUIRE(x && gamma && beta && y, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && F > 0 && ldx >= F && ldy >= F, SMX_E_SHAPE);
    SMX_REQUIRE(F <= 64 * LN_MAXC, SMX_E_UNSUPPORTED);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, smx_s(stream), x, (long)ldx,
                       (long)rows, F, gamma, beta, eps, y, (long)ldy, mean, rstd);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int64_t smx_layernorm_backward_ws_floats(int64_t rows, int32_t F) {
    return rows > 0 && F > 0 ? 2 * ((rows + 15) / 16) * (int64_t)F : 0;
}

extern "C" int smx_layernorm_backward_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                          const float* rstd, const float* gamma, int64_t rows, int32_t F, int32_t relu_mask,
                                          float* dx, int64_t lddx, float* dgamma, float* dbeta, float* ws, int64_t ws_floats,
                                          smx_stream_t stream) {
    SMX_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && ws, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && F > 0 && lddy >= F && ldx >= F && lddx >= F, SMX_E_SHAPE);
    SMX_REQUIRE(F <= 64 * LN_MAXC, SMX_E_UNSUPPORTED);
    SMX_REQUIRE(ws_floats >= smx_layernorm_backward_ws_floats(rows, F), SMX_E_WORKSPACE);
    const int nblk = (int)((rows + 15) / 16);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, smx_s(stream), dy, (long)lddy, x, (long)ldx, mean,
                       rstd, gamma, (long)rows, F, relu_mask, dx, (long)lddx, ws);
    hipLaunchKernelGGL(layernorm_pgrad_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, smx_s(stream), ws, nblk, F,
                       dgamma, dbeta);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
