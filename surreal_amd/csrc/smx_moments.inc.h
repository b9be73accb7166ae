// Statistics reductions shared by smx_ppo.hip (their own launches) and smx_scan.hip (the one-launch
// learn epilogue).  Included at file scope.
#pragma once

// mergeable moments (count, mean, M2) of two disjoint sets (Chan et al.)
struct Mom { double n, mean, m2; };
static __device__ __forceinline__ Mom mom_merge(const Mom& a, const Mom& b) {
    if (b.n <= 0.0) return a;
    if (a.n <= 0.0) return b;
    Mom r;
    r.n = a.n + b.n;
    const double d = b.mean - a.mean;
    r.m2 = a.m2 + b.m2 + d * d * a.n * b.n / r.n;
    r.mean = a.mean + d * b.n / r.n;
    return r;
}
static __device__ __forceinline__ double shfl_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

// one wave: lane l folds the partial rows l, l + 64, ... of ONE epoch in order, then the 64 lanes are
// merged pairwise in a fixed tree (lane l with l + 32, 16, ...: the same order on every run)
static __device__ __forceinline__ void value_finalize_wave(const float* __restrict__ P, int nblk,
                                                           float* __restrict__ stats_e, int lane) {
    Mom d = {0.0, 0.0, 0.0}, g = {0.0, 0.0, 0.0};
    double sq = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        const float4 lo = *(const float4*)(P + 8 * b);
        const float4 hi = *(const float4*)(P + 8 * b + 4);
        const Mom db = {(double)lo.x, (double)lo.y, (double)lo.z}, gb = {(double)lo.x, (double)lo.w, (double)hi.x};
        d = mom_merge(d, db);
        g = mom_merge(g, gb);
        if (lo.x > 0.f) sq += (double)hi.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Mom od, og;
        od.n = shfl_d(d.n, lane + off); od.mean = shfl_d(d.mean, lane + off); od.m2 = shfl_d(d.m2, lane + off);
        og.n = od.n; og.mean = shfl_d(g.mean, lane + off); og.m2 = shfl_d(g.m2, lane + off);
        const double osq = shfl_d(sq, lane + off);
        if (lane < off) {
            d = mom_merge(d, od);
            g = mom_merge(g, og);
            sq += osq;
        }
    }
    if (lane == 0) {
        const double n = d.n;
        stats_e[SMX_VS_LOSS] = (float)(sq / n);          // ppo.py:326
        // 1 - var(returns - values) / var(returns), unbiased variances (ppo.py:325)
        stats_e[SMX_VS_EXPVAR] = 1.0f - (float)(d.m2 / (n - 1.0)) / (float)(g.m2 / (n - 1.0));
    }
}

// out[0] = mean(log_var); out[1..3] = mean over the D features of running_sum/count,
// running_sumsq/count and sqrt(running_sumsq/count - (running_sum/count)^2)  (no clamp: z_filter.py:90-98);
// one workgroup of blockDim.x threads (a multiple of 64, <= 1024); red: 16 x 4 doubles of LDS
static __device__ __forceinline__ void final_stats_block(const float* __restrict__ log_var, int A,
                                                         const float* __restrict__ rsum,
                                                         const float* __restrict__ rsumsq,
                                                         const float* __restrict__ count, int D,
                                                         float* __restrict__ out, double (*red)[4]) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int a = threadIdx.x; a < A; a += blockDim.x) s0 += (double)log_var[a];
    if (rsum) {
        const float cnt = __builtin_nontemporal_load(count);
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            const float m = __builtin_nontemporal_load(rsum + d) / cnt, q = __builtin_nontemporal_load(rsumsq + d) / cnt;
            s1 += (double)m;
            s2 += (double)q;
            s3 += (double)sqrtf(q - m * m);
        }
    }
    s0 = smx_wave_sum_d(s0); s1 = smx_wave_sum_d(s1); s2 = smx_wave_sum_d(s2); s3 = smx_wave_sum_d(s3);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w][0] = s0; red[w][1] = s1; red[w][2] = s2; red[w][3] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += red[i][threadIdx.x];
        out[threadIdx.x] = (float)(t / (double)(threadIdx.x == 0 ? A : (D > 0 ? D : 1)));
    }
}
