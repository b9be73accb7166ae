// The statistics of a DDPG iteration (surreal/learner/ddpg.py:335-342) as ONE workgroup's work, shared by the launch of its
// own (smx_ddpg.hip: ddpg_stats_kernel, 1024 threads) and by the row schedule's last launch, which runs it as an extra
// workgroup (smx_ddpg_rows.hip, 512 threads).  Included inside an anonymous namespace, after smx_common.h.
//   stats[6] = {actor_loss, critic_loss, action_norm, rewards, Q_target, Q_policy}: means over the rows, summed in double
//   (per thread over its rows, over the wavefront, over the wavefronts in order -- the thread count only moves additions
//   between those three levels: differences at double rounding, far below the float the mean is stored as);
//   stats[6] = max |action| (the reference asserts |a| <= 1 with two host syncs, ddpg.py:262-263); NaN if any action is.
//   mirror: a second destination for the same seven words (host-mapped memory: the learner reads them without a copy launch).
#pragma once

template <int NT>
__device__ __forceinline__ void ddpg_stats_block(const float* __restrict__ q, const float* __restrict__ y,
                                                 const float* __restrict__ rewards, const float* __restrict__ actions,
                                                 int ld_act, int A, const float* __restrict__ q_actor, long rows,
                                                 float* __restrict__ stats, float* __restrict__ mirror = nullptr) {
    constexpr int NW = NT / 64;
    __shared__ double red[NW];
    __shared__ float mx[NW];
    __shared__ int nanw[NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    float amax = 0.f;
    int anan = 0;
    for (long r = threadIdx.x; r < rows; r += NT) {
        const float d = q[r] - y[r];
        float nn = 0.f;
        for (int j = 0; j < A; ++j) {
            const float a = actions[r * ld_act + j];
            nn += a * a;
            if (a == a) amax = fmaxf(amax, fabsf(a));
            else anan = 1;                                    // a NaN action must fail the check
        }
        acc[0] += (double)(-q_actor[r]);
        acc[1] += (double)(d * d);
        acc[2] += (double)sqrtf(nn);
        acc[3] += (double)rewards[r];
        acc[4] += (double)y[r];
        acc[5] += (double)q[r];
    }
    for (int k = 0; k < 6; ++k) {
        double s = smx_wave_sum_d(acc[k]);
        __syncthreads();
        if (lane == 0) red[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < NW; ++i) t += red[i];
            stats[k] = (float)(t / (double)rows);
            if (mirror) mirror[k] = stats[k];
        }
    }
    // (round 5: a wavefront-level reduction -- thread 0 walking 1024 LDS words took 18 of the launch's 22 us)
    int bad = anan;
    float mxv = amax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mxv = fmaxf(mxv, __shfl_xor(mxv, off, 64));
        bad |= __shfl_xor(bad, off, 64);
    }
    if (lane == 0) { mx[w] = mxv; nanw[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        int b = 0;
        for (int i = 0; i < NW; ++i) { t = fmaxf(t, mx[i]); b |= nanw[i]; }
        stats[6] = b ? NAN : t;
        if (mirror) mirror[6] = stats[6];
    }
}
