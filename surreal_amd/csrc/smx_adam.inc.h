// clip_grad_norm_ + torch.optim.Adam's single-tensor step, shared by the optimiser launch (smx_ppo.hip:
// clip_adam_kernel) and by the weight-gradient launch that steps its own tiles (smx_gemm.hip: gemm32_adam_kernel) --
// one source, so that both form the same bits.  Included inside an anonymous namespace, after smx_epoch_pack.inc.h.
//   torch/nn/utils/clip_grad.py: clip_coef = max_norm / (total_norm + 1e-6), clamped to <= 1, always multiplied in
//   torch/optim/adam.py _single_tensor_adam (the reference's optimisers: surreal/learner/ppo.py:120-135)
#pragma once

struct AdamPack {              // optional: the fused epoch kernels' packed copy of the group's MLP, kept current by the step
    float* packed;
    long oW1, oW2, oW3;        // offsets of the three weight matrices inside theta
    int D, H1, H2, OUT;
};

struct AdamCoef {
    float coef, wd, w1, b2f, w2, neg_step_size, bc2_sqrt;
};

// the part of the coefficients that does not depend on the gradient norm (two double-precision pow(): formed while a
// caller waits for the norm)
__device__ __forceinline__ AdamCoef adam_coef_pre(const smx_ppo_ctrl_t& C, int which) {
    AdamCoef K;
    K.coef = 1.0f;
    const double beta1 = 0.9, beta2 = 0.999;
    const int step = which ? C.adam_step_critic : C.adam_step_actor;
    const double lr = (double)(which ? C.lr_critic : C.lr_actor);
    K.wd = which ? C.critic_weight_decay : C.actor_weight_decay;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    K.neg_step_size = (float)(-(lr / bc1));
    K.bc2_sqrt = (float)sqrt(bc2);
    K.w1 = (float)(1.0 - beta1);
    K.b2f = (float)beta2;
    K.w2 = (float)(1.0 - beta2);
    return K;
}
__device__ __forceinline__ float clip_coef(const smx_ppo_ctrl_t& C, int which, float norm) {
    const float max_norm = which ? C.critic_max_norm : C.actor_max_norm;
    float coef = 1.0f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
    return coef;
}
__device__ __forceinline__ AdamCoef adam_coef(const smx_ppo_ctrl_t& C, int which, float norm) {
    AdamCoef K = adam_coef_pre(C, which);
    K.coef = clip_coef(C, which, norm);
    return K;
}

// one element: g, p, exp_avg, exp_avg_sq -> the new p, exp_avg, exp_avg_sq
__device__ __forceinline__ void adam_step_core(const AdamCoef& K, float g, float p, float& mi, float& vi, float& pn) {
    const float eps = 1e-8f;
    g = g * K.coef;
    if (K.wd != 0.f) g = g + K.wd * p;                       // grad.add(param, alpha=wd)
    mi = mi + K.w1 * (g - mi);                               // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * K.b2f + K.w2 * (g * g);                        // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / K.bc2_sqrt + eps;
    pn = p + (K.neg_step_size * mi) / denom;                 // addcdiv_(exp_avg, denom, -step_size)
}

__device__ __forceinline__ void adam_step_one(const AdamCoef& K, const AdamPack& Q, float* __restrict__ theta,
                                              float* __restrict__ m, float* __restrict__ v, long i, float g, float p,
                                              float mi, float vi) {
    float pn;
    adam_step_core(K, g, p, mi, vi, pn);
    theta[i] = pn;
    m[i] = mi;
    v[i] = vi;
    if (Q.packed) {            // the same value into the forward / backward kernels' fragment-order copies
        float* Pk = Q.packed;
        if (i >= Q.oW1 && i < Q.oW1 + (long)Q.H1 * Q.D) {
            const int mm = (int)((i - Q.oW1) / Q.D), kk = (int)((i - Q.oW1) - (long)mm * Q.D);
            Pk[pack_pos(Q.D, mm, kk)] = pn;
        } else if (i >= Q.oW2 && i < Q.oW2 + (long)Q.H2 * Q.H1) {
            const int mm = (int)((i - Q.oW2) / Q.H1), kk = (int)((i - Q.oW2) - (long)mm * Q.H1);
            Pk[4 * pack_off(Q.D, Q.H1, Q.H2, Q.OUT, 1) + pack_pos(Q.H1, mm, kk)] = pn;
            Pk[4 * pack_off(Q.D, Q.H1, Q.H2, Q.OUT, 3) + pack_pos(Q.H2, kk, mm)] = pn;
        } else if (i >= Q.oW3 && i < Q.oW3 + (long)Q.OUT * Q.H2) {
            const int mm = (int)((i - Q.oW3) / Q.H2), kk = (int)((i - Q.oW3) - (long)mm * Q.H2);
            Pk[4 * pack_off(Q.D, Q.H1, Q.H2, Q.OUT, 2) + pack_pos(Q.H2, mm, kk)] = pn;
            Pk[4 * pack_off(Q.D, Q.H1, Q.H2, Q.OUT, 4) + pack_pos(Q.OUT, kk, mm)] = pn;
        }
    }
}
