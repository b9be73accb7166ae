// Windowed GAE / n-step returns, batch moments, advantage normalisation, z-filter.
// HBM-bound elementwise / reduction kernels: one wavefront per sub-trajectory row,
// coalesced row loads, LDS-staged masked values, wave-shuffle reductions.
// Reference: surreal/learner/ppo.py:387-418, surreal/model/z_filter.py:44-79.
#include "smx_common.h"
#include "smx_moments.inc.h"

// ---------------------------------------------------------------------------
// One wave per sub-trajectory b.  LDS per wave: Vm[N+1] | r[N] | delta[N]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gae_body(
    const float* __restrict__ values, const float* __restrict__ values_tail,
    const float* __restrict__ rewards,
    const float* __restrict__ dones, const float* __restrict__ gpow,
    const float* __restrict__ lpow, float gamma, float gamma_H, int B, int N, int H,
    float* __restrict__ adv, float* __restrict__ ret) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    const int per = 3 * N + 1;
    float* Vm = lds + (size_t)w * per;
    float* R = Vm + (N + 1);
    float* Dl = R + N;
    if (b < B) {
        // values [B, N+1], or [B, N] + values_tail [B] (bootstrap value of obs_next)
        const float* v = values + (size_t)b * (values_tail ? N : N + 1);
        const float* r = rewards + (size_t)b * N;
        const float* d = dones + (size_t)b * N;
        // (four 64-step slices at a time: every load of a slice first, unconditionally and from an index valid in every
        // lane, the conditions applied afterwards -- with `cond ? load : ...` per step hipcc waited for each load where it
        // was issued: two dependent memory round trips per slice in front of a 16 us launch)
        const float vt = values_tail ? values_tail[b] : 0.f;
        const int vlast = values_tail ? N - 1 : N;
        for (int t0 = 0; t0 <= N; t0 += 256) {
            float xv[4], dv[4], rv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = t0 + lane + 64 * i;
                xv[i] = v[t < vlast ? t : vlast];
                dv[i] = d[t >= 1 ? (t - 1 < N ? t - 1 : N - 1) : 0];
                rv[i] = r[t < N ? t : N - 1];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = t0 + lane + 64 * i;
                if (t <= N) {
                    float x = (values_tail && t == N) ? vt : xv[i];
                    if (t >= 1) x = x * (1.0f - dv[i]);  // values[:, 1:] *= 1 - dones   (ppo.py:387)
                    Vm[t] = x;
                    if (t < N) R[t] = rv[i];
                }
            }
        }
    }
    __syncthreads();
    if (b < B) {
        for (int t = lane; t < N; t += 64) {
            // tds = rewards + gamma * values[:, 1:] - values[:, :-1]   (ppo.py:390,410)
            float gv = gamma * Vm[t + 1];
            float s = R[t] + gv;
            Dl[t] = s - Vm[t];
        }
    }
    __syncthreads();
    if (b < B) {
    const int E = N - H + 1;
    if (E == 1) {
        // non-RNN: one return / advantage per row, a gamma-lambda weighted reduction
        float rs = 0.f, as = 0.f;
        for (int k = lane; k < N; k += 64) {
            rs += gpow[k] * R[k];
            as += (Dl[k] * gpow[k]) * lpow[k];
        }
        rs = smx_wave_sum(rs);
        as = smx_wave_sum(as);
        if (lane == 0) {
            ret[b] = rs + Vm[N] * gamma_H;  // ppo.py:409
            adv[b] = as;                    // ppo.py:411
        }
    } else {
        // RNN branch: E sliding windows of H terms  (ppo.py:397-400)
        for (int s = lane; s < E; s += 64) {
            float rs = 0.f, as = 0.f;
            for (int k = 0; k < H; ++k) {
                rs += gpow[k] * R[s + k];
                as += (Dl[s + k] * gpow[k]) * lpow[k];
            }
            ret[(size_t)b * E + s] = rs + Vm[s + H] * gamma_H;
            adv[(size_t)b * E + s] = as;
        }
    }
    }
}

__global__ __launch_bounds__(256) void gae_kernel(
    const float* __restrict__ values, const float* __restrict__ values_tail,
    const float* __restrict__ rewards, const float* __restrict__ dones, const float* __restrict__ gpow,
    const float* __restrict__ lpow, float gamma, float gamma_H, int B, int N, int H,
    float* __restrict__ adv, float* __restrict__ ret) {
    gae_body(values, values_tail, rewards, dones, gpow, lpow, gamma, gamma_H, B, N, H, adv, ret);
}

// the LAST workgroup of a launch to get here returns true (device-scope release / acquire around a
// ticket counter, which it resets for the next launch): the place for a launch's global epilogue
__device__ __forceinline__ bool last_block_done(int* ticket) {
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int t = atomicAdd(ticket, 1);
        is_last = (t == (int)gridDim.x - 1);
        if (is_last) *ticket = 0;
        __threadfence();
    }
    __syncthreads();
    return is_last != 0;
}

// batch moments {n, mean, M2} of x[0, n) in fp64 (two passes) by ONE workgroup of blockDim.x <= 1024 threads
__device__ __forceinline__ void block_moments(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ double red[16];
    __shared__ double bc;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += (double)__builtin_nontemporal_load(x + i);
    s = smx_wave_sum_d(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += red[i];
        bc = t / (double)n;
    }
    __syncthreads();
    const double mean = bc;
    double q = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const double d = (double)__builtin_nontemporal_load(x + i) - mean;
        q += d * d;
    }
    q = smx_wave_sum_d(q);
    __syncthreads();
    if (lane == 0) red[w] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += red[i];
        out[0] = (float)n;
        out[1] = (float)mean;
        out[2] = (float)t;
        bc = t;
    }
    __syncthreads();
}

// GAE + (optionally) the batch normalisation of the advantages (ppo.py:402-405, 413-416) in the same
// launch: the last workgroup to finish forms the moments and normalises in place
__global__ __launch_bounds__(256) void gae_norm_kernel(
    const float* __restrict__ values, const float* __restrict__ values_tail,
    const float* __restrict__ rewards, const float* __restrict__ dones, const float* __restrict__ gpow,
    const float* __restrict__ lpow, float gamma, float gamma_H, int B, int N, int H,
    float* __restrict__ adv, float* __restrict__ ret, float* __restrict__ norm_mom, float min_std,
    int* __restrict__ ticket) {
    gae_body(values, values_tail, rewards, dones, gpow, lpow, gamma, gamma_H, B, N, H, adv, ret);
    if (!last_block_done(ticket)) return;
    const long n = (long)B * (N - H + 1);
    block_moments(adv, n, norm_mom);
    const float cnt = (float)n, mean = norm_mom[1], m2 = norm_mom[2];
    const float stdv = sqrtf(m2 / (cnt - 1.0f));             // advs.std(): unbiased
    const float den = (min_std > stdv) ? min_std : stdv;     // Python max(std, 1e-4)
    for (long i = threadIdx.x; i < n; i += blockDim.x) adv[i] = (__builtin_nontemporal_load(adv + i) - mean) / den;
}

extern "C" int smx_windowed_gae_returns_f32(const float* values, const float* values_tail,
                                            const float* rewards,
                                            const float* dones, const float* gamma_pow,
                                            const float* lam_pow, float gamma, float gamma_H,
                                            int32_t B, int32_t N, int32_t H, float* adv,
                                            float* ret, smx_stream_t stream) {
    SMX_REQUIRE(values && rewards && dones && gamma_pow && lam_pow && adv && ret, SMX_E_NULL);
    SMX_REQUIRE(B > 0 && N > 0 && H > 0 && H <= N, SMX_E_SHAPE);
    const size_t lds = (size_t)4 * (3 * (size_t)N + 1) * sizeof(float);
    SMX_REQUIRE(lds <= 160 * 1024, SMX_E_UNSUPPORTED);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gae_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(gae_kernel, dim3((B + 3) / 4), dim3(256), lds, smx_s(stream), values,
                       values_tail, rewards, dones, gamma_pow, lam_pow, gamma, gamma_H, B, N, H, adv, ret);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_windowed_gae_norm_f32(const float* values, const float* values_tail, const float* rewards,
                                         const float* dones, const float* gamma_pow, const float* lam_pow,
                                         float gamma, float gamma_H, int32_t B, int32_t N, int32_t H, float* adv,
                                         float* ret, float* adv_moments, float min_std, int32_t* ticket,
                                         smx_stream_t stream) {
    SMX_REQUIRE(values && rewards && dones && gamma_pow && lam_pow && adv && ret && adv_moments && ticket, SMX_E_NULL);
    SMX_REQUIRE(B > 0 && N > 0 && H > 0 && H <= N, SMX_E_SHAPE);
    const size_t lds = (size_t)4 * (3 * (size_t)N + 1) * sizeof(float);
    SMX_REQUIRE(lds <= 128 * 1024, SMX_E_UNSUPPORTED);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gae_norm_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(gae_norm_kernel, dim3((B + 3) / 4), dim3(256), lds, smx_s(stream), values, values_tail,
                       rewards, dones, gamma_pow, lam_pow, gamma, gamma_H, B, N, H, adv, ret, adv_moments, min_std,
                       (int*)ticket);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// ---------------------------------------------------------------------------
// Reward scale + RewardFilter (ppo.py:452-455, reward_filter.py:33-57) in ONE launch: x = r * scale;
// out = clamp((x - mean) / std, -5, 5) with mean / std from the statistics BEFORE this batch; then the
// statistics take the batch in: count += n, running_sum += sum(x), running_sumsq = sum(x * x) (assigned,
// not accumulated: reward_filter.py:42).  Block partial sums in fp64, added in block order by the last
// workgroup to finish (no floating-point atomics: the result does not depend on the schedule).
// ---------------------------------------------------------------------------
constexpr int RF_MAX_BLOCKS = 64;

__global__ __launch_bounds__(256) void reward_filter_kernel(const float* __restrict__ r, long n, float scale, int filter,
                                                            float* __restrict__ state, float eps, int update,
                                                            float* __restrict__ out, float* __restrict__ sums,
                                                            double* __restrict__ partials, int* __restrict__ ticket) {
    __shared__ double red[8];
    float mean = 0.f, sd = 1.f;
    if (filter) {
        const float cnt = state[0];
        mean = state[1] / cnt;
        sd = sqrtf(state[2] / cnt - mean * mean);          // pow(0.5): NaN for a negative argument
        if (sd == sd) sd = fmaxf(sd, eps);                   // torch.clamp keeps NaN
    }
    double s1 = 0.0, s2 = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = r[i] * scale;
        float v = x;
        if (filter) {
            v = (x - mean) / sd;
            if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);
        }
        out[i] = v;
        s1 += (double)x;
        s2 += (double)x * (double)x;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    s1 = smx_wave_sum_d(s1);
    s2 = smx_wave_sum_d(s2);
    if (lane == 0) { red[w] = s1; red[4 + w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        partials[2 * blockIdx.x + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
    if (!last_block_done(ticket)) return;
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b) {
            t1 += __builtin_nontemporal_load(partials + 2 * b);
            t2 += __builtin_nontemporal_load(partials + 2 * b + 1);
        }
        if (sums) { sums[0] = (float)n; sums[1] = (float)t1; sums[2] = (float)t2; }
        if (update) {
            state[0] += (float)n;
            state[1] += (float)t1;
            state[2] = (float)t2;
        }
    }
}

extern "C" int smx_reward_filter_f32(const float* rewards, int64_t n, float scale, int32_t use_filter, float* state,
                                     float eps, int32_t update_state, float* out, float* sums, double* partials,
                                     int32_t* ticket, smx_stream_t stream) {
    SMX_REQUIRE(rewards && out && partials && ticket, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    SMX_REQUIRE(!(use_filter || update_state) || state, SMX_E_NULL);
    SMX_REQUIRE(((uintptr_t)partials & 7) == 0, SMX_E_ALIGN);
    long blocks = (n + 1023) / 1024;
    if (blocks > RF_MAX_BLOCKS) blocks = RF_MAX_BLOCKS;
    hipLaunchKernelGGL(reward_filter_kernel, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), rewards, (long)n, scale,
                       use_filter, state, eps, update_state, out, sums, partials, (int*)ticket);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int32_t smx_reward_filter_partials(void) { return 2 * RF_MAX_BLOCKS; }

// ---------------------------------------------------------------------------
// moments: {n, mean, M2}; two passes in fp64 inside one workgroup (n is B*E <= ~1e5)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void moments_kernel(const float* __restrict__ x, long n,
                                                       float* __restrict__ out) {
    __shared__ double red[16];
    __shared__ double bc;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) s += (double)x[i];
    s = smx_wave_sum_d(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        bc = t / (double)n;
    }
    __syncthreads();
    const double mean = bc;
    double q = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        double d = (double)x[i] - mean;
        q += d * d;
    }
    q = smx_wave_sum_d(q);
    __syncthreads();
    if (lane == 0) red[w] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = (float)n;
        out[1] = (float)mean;
        out[2] = (float)t;
    }
}

extern "C" int smx_moments_f32(const float* x, int64_t n, float* moments, smx_stream_t stream) {
    SMX_REQUIRE(x && moments, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(moments_kernel, dim3(1), dim3(1024), 0, smx_s(stream), x, (long)n, moments);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// Chan et al. pairwise merge of per-rank (n, mean, M2)
__global__ void moments_merge_kernel(const float* __restrict__ parts, int k,
                                     float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int i = 0; i < k; ++i) {
        const double nb = parts[3 * i], mb = parts[3 * i + 1], qb = parts[3 * i + 2];
        if (nb <= 0.0) continue;
        const double nt = n + nb, d = mb - mean;
        m2 = m2 + qb + d * d * n * nb / nt;
        mean = mean + d * nb / nt;
        n = nt;
    }
    out[0] = (float)n;
    out[1] = (float)mean;
    out[2] = (float)m2;
}

extern "C" int smx_moments_merge_f32(const float* parts, int32_t k, float* out,
                                     smx_stream_t stream) {
    SMX_REQUIRE(parts && out, SMX_E_NULL);
    SMX_REQUIRE(k > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(moments_merge_kernel, dim3(1), dim3(64), 0, smx_s(stream), parts, k, out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

__global__ __launch_bounds__(256) void adv_normalize_kernel(float* __restrict__ x, long n,
                                                            const float* __restrict__ mom,
                                                            float min_std) {
    const float cnt = mom[0], mean = mom[1], m2 = mom[2];
    // advs.std(): unbiased; n == 1 gives 0/0 = NaN exactly as torch does
    const float stdv = sqrtf(m2 / (cnt - 1.0f));
    // Python max(std, 1e-4): returns 1e-4 only when 1e-4 > std (NaN std propagates)
    const float den = (min_std > stdv) ? min_std : stdv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        x[i] = (x[i] - mean) / den;
}

extern "C" int smx_adv_normalize_f32(float* x, int64_t n, const float* moments, float min_std,
                                     smx_stream_t stream) {
    SMX_REQUIRE(x && moments, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adv_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), x,
                       (long)n, moments, min_std);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// ---------------------------------------------------------------------------
// z-filter
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zstats_kernel(const float* __restrict__ rs,
                                                     const float* __restrict__ rsq,
                                                     const float* __restrict__ cnt, int D,
                                                     float eps, float* __restrict__ mean,
                                                     float* __restrict__ stdv) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= D) return;
    const float c = cnt[0];
    const float m = rs[k] / c;                // z_filter.py:74
    const float var = rsq[k] / c - m * m;     // z_filter.py:75
    float s = sqrtf(var);                     // .pow(0.5): NaN for var < 0, like torch
    if (s == s) s = fmaxf(s, eps);            // torch.clamp(min=eps) propagates NaN
    mean[k] = m;
    stdv[k] = s;
}

extern "C" int smx_zfilter_stats_f32(const float* running_sum, const float* running_sumsq,
                                     const float* count, int32_t D, float eps, float* mean_out,
                                     float* std_out, smx_stream_t stream) {
    SMX_REQUIRE(running_sum && running_sumsq && count && mean_out && std_out, SMX_E_NULL);
    SMX_REQUIRE(D > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(zstats_kernel, dim3((D + 255) / 256), dim3(256), 0, smx_s(stream),
                       running_sum, running_sumsq, count, D, eps, mean_out, std_out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

static __device__ __forceinline__ float zclamp(float x, float m, float s) {
    float v = (x - m) / s;   // z_filter.py:77
    // torch.clamp(v, -5, 5): NaN stays NaN
    if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);
    return v;
}

__global__ __launch_bounds__(256) void zforward_kernel(const float* __restrict__ x, long ldx,
                                                       long total, int D,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ stdv,
                                                       float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / D;
        const int k = (int)(i - r * D);
        out[i] = zclamp(x[r * ldx + k], mean[k], stdv[k]);
    }
}

extern "C" int smx_zfilter_forward_f32(const float* x, int64_t ldx, int64_t rows, int32_t D,
                                       const float* mean, const float* stdv, float* out,
                                       smx_stream_t stream) {
    SMX_REQUIRE(x && mean && stdv && out, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && D > 0 && ldx >= D, SMX_E_SHAPE);
    const long total = (long)rows * D;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zforward_kernel, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), x,
                       (long)ldx, total, D, mean, stdv, out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// the same straight from the running sums (what ZFilter.forward does on every call,
// z_filter.py:74-77): an acting agent applies the filter to one observation per step, where a
// separate statistics launch would double the cost
__global__ __launch_bounds__(256) void zforward_sums_kernel(const float* __restrict__ x, long ldx,
                                                            long total, int D,
                                                            const float* __restrict__ rs,
                                                            const float* __restrict__ rsq,
                                                            const float* __restrict__ cnt, float eps,
                                                            float* __restrict__ out) {
    const float c = cnt[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / D;
        const int k = (int)(i - r * D);
        const float m = rs[k] / c;
        const float var = rsq[k] / c - m * m;
        float sd = sqrtf(var);
        if (sd == sd) sd = fmaxf(sd, eps);
        out[i] = zclamp(x[r * ldx + k], m, sd);
    }
}

extern "C" int smx_zfilter_forward_sums_f32(const float* x, int64_t ldx, int64_t rows, int32_t D,
                                            const float* running_sum, const float* running_sumsq,
                                            const float* count, float eps, float* out,
                                            smx_stream_t stream) {
    SMX_REQUIRE(x && running_sum && running_sumsq && count && out, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && D > 0 && ldx >= D, SMX_E_SHAPE);
    const long total = (long)rows * D;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zforward_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), x,
                       (long)ldx, total, D, running_sum, running_sumsq, count, eps, out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// Column sums of x and x*x: block = 64 columns x 16 row-lanes (1024 threads); each thread walks
// rows r = g, g+16, ... (the rows of obs[:, 0, :] are N*D floats apart: latency-bound, so depth
// comes from 16 row-lanes x 4 independent loads in flight), then a fixed-order LDS reduction.
__global__ __launch_bounds__(1024) void zupdate_kernel(const float* __restrict__ x, long ldx,
                                                       long rows, int D, float* __restrict__ rs,
                                                       float* __restrict__ rsq,
                                                       float* __restrict__ cnt, float count_rows) {
    __shared__ float s1[16][64], s2[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if (col < D) {
        long r = g;
        for (; r + 48 < rows; r += 64) {
            const float v0 = x[r * ldx + col], v1 = x[(r + 16) * ldx + col];
            const float v2 = x[(r + 32) * ldx + col], v3 = x[(r + 48) * ldx + col];
            a0 += v0; q0 += v0 * v0;
            a1 += v1; q1 += v1 * v1;
            a2 += v2; q2 += v2 * v2;
            a3 += v3; q3 += v3 * v3;
        }
        for (; r < rows; r += 16) {
            const float v = x[r * ldx + col];
            a0 += v; q0 += v * v;
        }
    }
    s1[g][c] = (a0 + a1) + (a2 + a3);
    s2[g][c] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    if (g == 0 && col < D) {
        float ta = 0.f, tq = 0.f;
        for (int k = 0; k < 16; ++k) { ta += s1[k][c]; tq += s2[k][c]; }
        rs[col] += ta;    // z_filter.py:55
        rsq[col] += tq;   // z_filter.py:56
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[0] += count_rows;  // z_filter.py:57
}

extern "C" int smx_zfilter_update_f32(const float* x, int64_t ldx, int64_t rows, int32_t D,
                                      float* running_sum, float* running_sumsq, float* count,
                                      float count_rows, smx_stream_t stream) {
    SMX_REQUIRE(x && running_sum && running_sumsq && count, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && D > 0 && ldx >= D, SMX_E_SHAPE);
    hipLaunchKernelGGL(zupdate_kernel, dim3((D + 63) / 64), dim3(1024), 0, smx_s(stream), x,
                       (long)ldx, (long)rows, D, running_sum, running_sumsq, count, count_rows);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}


// The same over MANY rows (the stems' z-update over B * E ~ 10^5 observation rows): zupdate_kernel is ONE workgroup per 64
// columns walking every row -- 0.99 ms at 126 976 x 17 (profiles/r05_lstm_1024x128_kernel_stats.csv).  Here the rows are cut
// into `chunks` pieces, workgroup (chunk, column block) leaves its column sums in the caller's scratch and a second small
// launch adds the chunks in order (deterministic) into the running sums.
__global__ __launch_bounds__(1024) void zupdate_part_kernel(const float* __restrict__ x, long ldx, long rows, int D, int chunks,
                                                            float* __restrict__ part) {
    __shared__ float s1[16][64], s2[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + c;
    const long per = (rows + chunks - 1) / chunks;
    const long r0 = (long)blockIdx.x * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f;
    if (col < D) {
        long r = r0 + g;
        for (; r + 16 < r1; r += 32) {
            const float v0 = x[r * ldx + col], v1 = x[(r + 16) * ldx + col];
            a0 += v0; q0 += v0 * v0;
            a1 += v1; q1 += v1 * v1;
        }
        if (r < r1) {
            const float v = x[r * ldx + col];
            a0 += v; q0 += v * v;
        }
    }
    s1[g][c] = a0 + a1;
    s2[g][c] = q0 + q1;
    __syncthreads();
    if (g == 0 && col < D) {
        float ta = 0.f, tq = 0.f;
        for (int k = 0; k < 16; ++k) { ta += s1[k][c]; tq += s2[k][c]; }
        part[((size_t)blockIdx.x * 2 + 0) * D + col] = ta;
        part[((size_t)blockIdx.x * 2 + 1) * D + col] = tq;
    }
}

__global__ __launch_bounds__(256) void zupdate_merge_kernel(const float* __restrict__ part, int chunks, int D,
                                                            float* __restrict__ rs, float* __restrict__ rsq,
                                                            float* __restrict__ cnt, float count_rows) {
    for (int col = blockIdx.x * 256 + threadIdx.x; col < D; col += gridDim.x * 256) {
        float ta = 0.f, tq = 0.f;
        for (int k = 0; k < chunks; ++k) {
            ta += part[((size_t)k * 2 + 0) * D + col];
            tq += part[((size_t)k * 2 + 1) * D + col];
        }
        rs[col] += ta;    // z_filter.py:55
        rsq[col] += tq;   // z_filter.py:56
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[0] += count_rows;  // z_filter.py:57
}

extern "C" int64_t smx_zfilter_update_ws_floats(int64_t rows, int32_t D) {
    if (rows < 16384 || D <= 0) return 0;           // few rows: the one-launch form
    const long chunks = rows / 512 < 256 ? rows / 512 : 256;
    return 2 * chunks * (int64_t)D;
}

extern "C" int smx_zfilter_update_ws_f32(const float* x, int64_t ldx, int64_t rows, int32_t D, float* running_sum,
                                         float* running_sumsq, float* count, float count_rows, float* ws, int64_t ws_floats,
                                         smx_stream_t stream) {
    const int64_t need = smx_zfilter_update_ws_floats(rows, D);
    if (need == 0 || !ws || ws_floats < need)
        return smx_zfilter_update_f32(x, ldx, rows, D, running_sum, running_sumsq, count, count_rows, stream);
    SMX_REQUIRE(x && running_sum && running_sumsq && count, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && D > 0 && ldx >= D, SMX_E_SHAPE);
    const int chunks = (int)(need / (2 * D));
    hipLaunchKernelGGL(zupdate_part_kernel, dim3((unsigned)chunks, (unsigned)((D + 63) / 64)), dim3(1024), 0, smx_s(stream), x,
                       (long)ldx, (long)rows, D, chunks, ws);
    hipLaunchKernelGGL(zupdate_merge_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, smx_s(stream), ws, chunks, D,
                       running_sum, running_sumsq, count, count_rows);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}


// ---------------------------------------------------------------------------
// What PPOLearner._optimize does after its epoch loops (ppo.py:565-584), in ONE launch instead of
// four: the explained variance / value loss of every value epoch from their block moments, the
// moments of the return targets, the z-filter update on the step-0 observations -- independent
// pieces, one per group of workgroups -- and, by the last workgroup to finish, the means the learner
// reports (they read the UPDATED z-filter sums).
//   blocks [0, nz)            column sums of 64 observation features each (zupdate_kernel's scheme)
//   block  nz                 moments of `ret`
//   blocks (nz, nz + nv]      value-loss finalize, 16 epochs (waves) per block
// ---------------------------------------------------------------------------
struct EpilogueArgs {
    const float* x; long ldx; long rows; int D;
    float *rs, *rsq, *cnt; float count_rows;
    const float* ret; long n_ret; float* ret_mom;
    const float* vpartials; int n_epochs, nblk; float* vstats; int vstride;
    const float* log_var; int A; float* out4;
    int* ticket;
    int nz;
};

__global__ __launch_bounds__(1024) void learn_epilogue_kernel(EpilogueArgs P) {
    __shared__ float s1[16][64], s2[16][64];
    __shared__ double red4[16][4];
    const int blk = blockIdx.x;
    if (blk < P.nz) {
        const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int col = blk * 64 + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        if (col < P.D) {
            const float* x = P.x;
            const long ldx = P.ldx, rows = P.rows;
            long r = g;
            // (sixteen loads in flight; the additions in the order of four passes of the loop below: bit-identical sums.
            // With four in flight a thread's 64 rows were sixteen dependent round trips -- 10 of this launch's 12 us)
            for (; r + 240 < rows; r += 256) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = x[(r + 16 * i) * ldx + col];
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    a0 += v[i]; q0 += v[i] * v[i];
                    a1 += v[i + 1]; q1 += v[i + 1] * v[i + 1];
                    a2 += v[i + 2]; q2 += v[i + 2] * v[i + 2];
                    a3 += v[i + 3]; q3 += v[i + 3] * v[i + 3];
                }
            }
            for (; r + 48 < rows; r += 64) {
                const float v0 = x[r * ldx + col], v1 = x[(r + 16) * ldx + col];
                const float v2 = x[(r + 32) * ldx + col], v3 = x[(r + 48) * ldx + col];
                a0 += v0; q0 += v0 * v0;
                a1 += v1; q1 += v1 * v1;
                a2 += v2; q2 += v2 * v2;
                a3 += v3; q3 += v3 * v3;
            }
            for (; r < rows; r += 16) {
                const float v = x[r * ldx + col];
                a0 += v; q0 += v * v;
            }
        }
        s1[g][c] = (a0 + a1) + (a2 + a3);
        s2[g][c] = (q0 + q1) + (q2 + q3);
        __syncthreads();
        if (g == 0 && col < P.D) {
            float ta = 0.f, tq = 0.f;
            for (int k = 0; k < 16; ++k) { ta += s1[k][c]; tq += s2[k][c]; }
            P.rs[col] += ta;    // z_filter.py:55
            P.rsq[col] += tq;   // z_filter.py:56
        }
        if (blk == 0 && threadIdx.x == 0) P.cnt[0] += P.count_rows;  // z_filter.py:57
    } else if (blk == P.nz) {
        block_moments(P.ret, P.n_ret, P.ret_mom);
    } else {
        const int e = (blk - P.nz - 1) * 16 + (threadIdx.x >> 6);
        if (e < P.n_epochs)
            value_finalize_wave(P.vpartials + (size_t)e * P.nblk * 8, P.nblk, P.vstats + (size_t)e * P.vstride,
                                threadIdx.x & 63);
    }
    if (!last_block_done(P.ticket)) return;
    final_stats_block(P.log_var, P.A, P.nz ? P.rs : nullptr, P.rsq, P.cnt, P.D, P.out4, red4);
}

extern "C" int smx_ppo_learn_epilogue_f32(const smx_learn_epilogue_t* a, smx_stream_t stream) {
    SMX_REQUIRE(a && a->ret && a->ret_moments && a->log_var && a->out4 && a->ticket, SMX_E_NULL);
    SMX_REQUIRE(a->n_ret > 0 && a->A > 0, SMX_E_SHAPE);
    SMX_REQUIRE(!a->x || (a->running_sum && a->running_sumsq && a->count && a->rows > 0 && a->D > 0 && a->ldx >= a->D),
                SMX_E_SHAPE);
    SMX_REQUIRE(a->n_epochs == 0 || (a->v_partials && a->v_stats && a->nblk > 0 && a->stats_stride >= 2), SMX_E_SHAPE);
    SMX_REQUIRE(a->n_epochs == 0 || ((uintptr_t)a->v_partials & 15) == 0, SMX_E_ALIGN);
    EpilogueArgs P;
    P.x = a->x; P.ldx = (long)a->ldx; P.rows = (long)a->rows; P.D = a->D;
    P.rs = a->running_sum; P.rsq = a->running_sumsq; P.cnt = a->count; P.count_rows = a->count_rows;
    P.ret = a->ret; P.n_ret = (long)a->n_ret; P.ret_mom = a->ret_moments;
    P.vpartials = a->v_partials; P.n_epochs = a->n_epochs; P.nblk = a->nblk; P.vstats = a->v_stats;
    P.vstride = a->stats_stride;
    P.log_var = a->log_var; P.A = a->A; P.out4 = a->out4; P.ticket = (int*)a->ticket;
    P.nz = a->x ? (a->D + 63) / 64 : 0;
    const int nv = (a->n_epochs + 15) / 16;
    hipLaunchKernelGGL(learn_epilogue_kernel, dim3(P.nz + 1 + nv), dim3(1024), 0, smx_s(stream), P);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_abi_version(void) { return 1; }

extern "C" const char* smx_error_string(int code) {
    switch (code) {
        case SMX_OK: return "ok";
        case SMX_E_NULL: return "required pointer is NULL";
        case SMX_E_SHAPE: return "non-positive or inconsistent dimension";
        case SMX_E_UNSUPPORTED: return "shape outside what the kernel is built for";
        case SMX_E_WORKSPACE: return "workspace too small";
        case SMX_E_ALIGN: return "pointer not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown smx error";
    }
}
