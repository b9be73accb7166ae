// Device-resident replay storage and sub-trajectory windowing (HBM-bound row copies).
// Reference: surreal/replay/fifo_replay.py:27-48, surreal/replay/uniform_replay.py:36-47,
// surreal/env/exp_sender_wrapper.py:209-264.  The reference keeps Python lists of experience
// dicts and pays pyarrow (de)serialisation around every insert/sample; here each experience field
// is one [capacity, width] fp32 table in HBM and insert/sample are coalesced row copies: one
// wavefront moves one row with 16-byte lanes when the width allows.
#include "smx_common.h"

namespace {

__device__ __forceinline__ float zclamp(float x, float m, float s) {
    float v = (x - m) / s;   // z_filter.py:77
    if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);
    return v;
}

// copy `n` rows of `width` floats: dst row i <- src row map(i).  A wavefront moves one SEGMENT of
// one row (ROW_SEG floats): a learner batch has few, very wide rows (1024 sub-trajectories of
// 192 KB) and one wave per row left 3/4 of the CUs idle with a single load in flight per lane
// (2.9 TB/s; 5.3 TB/s = the device's copy rate with segments and four loads in flight).
constexpr int ROW_SEG = 2048;             // floats = 8 KB per segment
constexpr int ROW_SEG_BYTES = ROW_SEG * 4;

// Rows are opaque byte strings: `gran` is the widest unit both row pitch and base addresses allow
// (16: float4 lanes, 4: dwords, 1: bytes) -- fp32 fields are rows of 4-byte units, uint8 camera frames
// (3 x 84 x 84 = 21 168 B = 1323 float4) move as what they are instead of being widened to fp32.
template <typename SrcRow, typename DstRow>
__device__ __forceinline__ void copy_rows_bytes_part(const unsigned char* __restrict__ src,
                                                     unsigned char* __restrict__ dst, long n, long row_bytes,
                                                     int gran, SrcRow srow, DstRow drow, long wave, long nwaves);

template <typename SrcRow, typename DstRow>
__device__ __forceinline__ void copy_rows_bytes(const unsigned char* __restrict__ src,
                                                unsigned char* __restrict__ dst, long n, long row_bytes,
                                                int gran, SrcRow srow, DstRow drow) {
    copy_rows_bytes_part(src, dst, n, row_bytes, gran, srow, drow, ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6,
                         ((long)gridDim.x * blockDim.x) >> 6);
}

// the same for the waves [wave, wave + nwaves, ...) of a PART of the grid (several tables gathered by one launch)
template <typename SrcRow, typename DstRow>
__device__ __forceinline__ void copy_rows_bytes_part(const unsigned char* __restrict__ src,
                                                     unsigned char* __restrict__ dst, long n, long row_bytes,
                                                     int gran, SrcRow srow, DstRow drow, long wave, long nwaves) {
    const int lane = threadIdx.x & 63;
    const int nseg = (int)((row_bytes + ROW_SEG_BYTES - 1) / ROW_SEG_BYTES);
    const long items = n * nseg;
    for (long it = wave; it < items; it += nwaves) {
        const long i = it / nseg;
        const long k0 = (it - i * nseg) * (long)ROW_SEG_BYTES;
        const int len = (int)min((long)ROW_SEG_BYTES, row_bytes - k0);
        const unsigned char* s = src + srow(i) * row_bytes + k0;
        unsigned char* d = dst + drow(i) * row_bytes + k0;
        if (gran == 16) {
            const float4* s4 = reinterpret_cast<const float4*>(s);
            float4* d4 = reinterpret_cast<float4*>(d);
            const int n4 = len >> 4;
            int k = lane;
            for (; k + 192 < n4; k += 256) {            // four independent 16-byte loads per lane
                const float4 a = s4[k], b = s4[k + 64], c = s4[k + 128], e = s4[k + 192];
                d4[k] = a; d4[k + 64] = b; d4[k + 128] = c; d4[k + 192] = e;
            }
            for (; k < n4; k += 64) d4[k] = s4[k];
        } else if (gran == 4) {
            const uint32_t* s1 = reinterpret_cast<const uint32_t*>(s);
            uint32_t* d1 = reinterpret_cast<uint32_t*>(d);
            for (int k = lane; k < (len >> 2); k += 64) d1[k] = s1[k];
        } else {
            for (int k = lane; k < len; k += 64) d[k] = s[k];
        }
    }
}

template <typename SrcRow, typename DstRow>
__device__ __forceinline__ void copy_rows(const float* __restrict__ src, float* __restrict__ dst,
                                          long n, int width, bool vec, SrcRow srow, DstRow drow) {
    copy_rows_bytes(reinterpret_cast<const unsigned char*>(src), reinterpret_cast<unsigned char*>(dst), n,
                    (long)width * 4, vec ? 16 : 4, srow, drow);
}

__global__ __launch_bounds__(256) void ring_insert_kernel(float* __restrict__ table, long capacity,
                                                          int width, long cursor,
                                                          const float* __restrict__ src, long n,
                                                          int vec) {
    copy_rows(src, table, n, width, vec != 0, [](long i) { return i; },
              [=](long i) { return (cursor + i) % capacity; });
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table,
                                                          long capacity, int width,
                                                          const int64_t* __restrict__ idx, long n,
                                                          float* __restrict__ dst, int vec) {
    copy_rows(table, dst, n, width, vec != 0,
              [=](long i) {
                  long j = (long)idx[i];
                  j = j < 0 ? 0 : (j >= capacity ? capacity - 1 : j);
                  return j;
              },
              [](long i) { return i; });
}

// Philox4x32-10 (Salmon et al. 2011), counter = (offset + i, 0), key = seed
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void uniform_indices_kernel(int64_t* __restrict__ idx, long n,
                                                              uint64_t len, uint64_t seed,
                                                              uint64_t offset) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t ctr = offset + (uint64_t)i;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    // 64 random bits -> [0, len) by 128-bit multiply-high (bias < len / 2^64)
    const uint64_t r64 = ((uint64_t)c[0] << 32) | c[1];
    idx[i] = (int64_t)__umul64hi(r64, len);
}

// the generator by itself, for arbitrary counters and keys: the known-answer self-test (Random123's kat_vectors)
__global__ __launch_bounds__(256) void philox_kat_kernel(const uint32_t* __restrict__ ck, long n, uint32_t* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t c[4] = {ck[6 * i], ck[6 * i + 1], ck[6 * i + 2], ck[6 * i + 3]};
    uint32_t k0 = ck[6 * i + 4], k1 = ck[6 * i + 5];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[4 * i] = c[0]; out[4 * i + 1] = c[1]; out[4 * i + 2] = c[2]; out[4 * i + 3] = c[3];
}

__device__ __forceinline__ long philox_index(uint64_t ctr, uint64_t len, uint64_t seed) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint64_t r64 = ((uint64_t)c[0] << 32) | c[1];
    return (long)__umul64hi(r64, len);
}

// A whole uniform sample in ONE launch (UniformReplay.sample, surreal/replay/uniform_replay.py:36-47): every field's
// table is gathered by its own part of the grid; the row indices are the ones smx_uniform_indices would write (the same
// Philox counters), formed where they are used -- or read from `idx` when the caller injects them.
struct GatherJobs {
    const unsigned char* table[8];
    unsigned char* dst[8];
    long row_bytes[8];
    int gran[8];
    int block0[9];          // first workgroup of each job, [n] = grid size
    int n;
};
__global__ __launch_bounds__(256) void uniform_gather_multi_kernel(GatherJobs J, long capacity, long rows,
                                                                   const int64_t* __restrict__ idx, uint64_t len,
                                                                   uint64_t seed, uint64_t offset,
                                                                   int64_t* __restrict__ idx_out) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) j += (k < J.n && (int)blockIdx.x >= J.block0[k]) ? 1 : 0;
    const long wave = ((long)((int)blockIdx.x - J.block0[j]) * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)(J.block0[j + 1] - J.block0[j]) * blockDim.x) >> 6;
    if (idx_out && j == 0)
        for (long i = (long)((int)blockIdx.x - J.block0[0]) * blockDim.x + threadIdx.x; i < rows;
             i += (long)(J.block0[1] - J.block0[0]) * blockDim.x)
            idx_out[i] = idx ? idx[i] : (int64_t)philox_index(offset + (uint64_t)i, len, seed);
    copy_rows_bytes_part(J.table[j], J.dst[j], rows, J.row_bytes[j], J.gran[j],
                         [=](long i) {
                             long r = idx ? (long)idx[i] : philox_index(offset + (uint64_t)i, len, seed);
                             return r < 0 ? 0 : (r >= capacity ? capacity - 1 : r);
                         },
                         [](long i) { return i; }, wave, nwaves);
}

__global__ __launch_bounds__(256) void window_emit_kernel(const float* __restrict__ src, int actors,
                                                          int T, int width, int start, int n_step,
                                                          int stride, int W, float* __restrict__ dst,
                                                          int vec) {
    const long n = (long)actors * W * n_step;
    copy_rows(src, dst, n, width, vec != 0,
              [=](long i) {
                  const long jw = i / n_step;
                  const int j = (int)(i - jw * n_step);
                  const long a = jw / W;
                  const int w = (int)(jw - a * W);
                  return a * T + start + (long)w * stride + j;
              },
              [](long i) { return i; });
}

__global__ __launch_bounds__(256) void ring_insert_bytes_kernel(unsigned char* __restrict__ table, long capacity,
                                                                long row_bytes, long cursor,
                                                                const unsigned char* __restrict__ src, long n,
                                                                int gran) {
    copy_rows_bytes(src, table, n, row_bytes, gran, [](long i) { return i; },
                    [=](long i) { return (cursor + i) % capacity; });
}

__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const unsigned char* __restrict__ table,
                                                                long capacity, long row_bytes,
                                                                const int64_t* __restrict__ idx, long n,
                                                                unsigned char* __restrict__ dst, int gran) {
    copy_rows_bytes(table, dst, n, row_bytes, gran,
                    [=](long i) {
                        long j = (long)idx[i];
                        j = j < 0 ? 0 : (j >= capacity ? capacity - 1 : j);
                        return j;
                    },
                    [](long i) { return i; });
}

__global__ __launch_bounds__(256) void window_emit_bytes_kernel(const unsigned char* __restrict__ src, int actors,
                                                                int T, long row_bytes, int start, int n_step,
                                                                int stride, int W, unsigned char* __restrict__ dst,
                                                                int gran) {
    const long n = (long)actors * W * n_step;
    copy_rows_bytes(src, dst, n, row_bytes, gran,
                    [=](long i) {
                        const long jw = i / n_step;
                        const int j = (int)(i - jw * n_step);
                        const long a = jw / W;
                        const int w = (int)(jw - a * W);
                        return a * T + start + (long)w * stride + j;
                    },
                    [](long i) { return i; });
}

// "obs stacking" (FrameStackWrapper, surreal/env/wrapper.py:407-472) as index arithmetic over the raw frames of a
// device-resident rollout (one frame per step, row 0 = the frame after reset): the observation of row s is the last
// n_stack frames on the channel axis, oldest first, and a reset fills the history with the first frame --
//   stacked(s)[i] = frames[max(s - (n_stack - 1) + i, first(s))]
// -- cut into W moving windows of n_step rows in the same gather (window_emit's arithmetic).
__global__ __launch_bounds__(256) void frame_stack_kernel(const unsigned char* __restrict__ src, int actors, int R,
                                                          long frame_bytes, int n_stack,
                                                          const int* __restrict__ first, int start, int n_step,
                                                          int stride, int W, unsigned char* __restrict__ dst, int gran) {
    const long n = (long)actors * W * n_step * n_stack;
    copy_rows_bytes(src, dst, n, frame_bytes, gran,
                    [=](long i) {
                        const long q = i / n_stack;
                        const int ii = (int)(i - q * n_stack);
                        const long jw = q / n_step;
                        const int j = (int)(q - jw * n_step);
                        const long a = jw / W;
                        const int w = (int)(jw - a * W);
                        const int sidx = start + w * stride + j;
                        const int lo = first ? first[a * R + sidx] : 0;
                        int f = sidx - (n_stack - 1) + ii;
                        f = f < lo ? lo : f;
                        return a * R + f;
                    },
                    [](long i) { return i; });
}

// the synthetic camera (surreal_amd.env.SyntheticEnv._frame): frame[c, y, x] = (37 c + 5 y + 11 x + 3 t +
// int(100 |s0|)) % 256 for every actor, s0 = the first state component the frame belongs to
__global__ __launch_bounds__(256) void synth_frame_kernel(const float* __restrict__ s0, long ld_s0, int n, int C, int H,
                                                          int Wd, int t, unsigned char* __restrict__ dst, long ld_dst) {
    const long per = (long)C * H * Wd;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)n * per) return;
    const long a = i / per;
    const int e = (int)(i - a * per);
    const int c = e / (H * Wd), yx = e - c * (H * Wd), y = yx / Wd, x = yx - y * Wd;
    const int shift = 3 * t + (int)(100.0 * (double)fabsf(s0[a * ld_s0]));
    dst[a * ld_dst + e] = (unsigned char)((37 * c + 5 * y + 11 * x + shift) % 256);
}

// one thread per (actor, k); the reward needs the whole action row: lanes k < 1 compute it
__global__ __launch_bounds__(256) void synth_env_step_kernel(
    float* __restrict__ state, const float* __restrict__ init_state,
    const float* __restrict__ actions, int n, int D, int A, int t, int episode_len, int slot, int T,
    float* __restrict__ obs_roll, float* __restrict__ act_roll, float* __restrict__ rew_roll,
    float* __restrict__ done_roll) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)n * D) return;
    const long a = i / D;
    const int k = (int)(i - a * D);
    const float s = state[i];
    float ac = actions[a * A + (k % A)];
    ac = fminf(fmaxf(ac, -1.0f), 1.0f);
    const float drift = 0.01f * (float)(((37 * k) % 17) - 8);
    float sn = (0.9f * s + 0.5f * ac) + drift;
    sn = fminf(fmaxf(sn, -10.0f), 10.0f);
    const bool done = (t + 1 >= episode_len);
    if (obs_roll) {
        obs_roll[(a * T + slot) * D + k] = s;
        // the observation AFTER this step (the terminal one when done): obs_next of a window
        // that ends here (exp_sender_wrapper.py:220-221)
        if (slot + 1 < T) obs_roll[(a * T + slot + 1) * D + k] = sn;
    }
    if (k < A && act_roll) {
        float av = actions[a * A + k];
        act_roll[(a * T + slot) * A + k] = fminf(fmaxf(av, -1.0f), 1.0f);
    }
    if (k == 0) {
        double q = 0.0;
        for (int j = 0; j < A; ++j) {
            float av = fminf(fmaxf(actions[a * A + j], -1.0f), 1.0f);
            q += (double)av * (double)av;
        }
        if (rew_roll) rew_roll[a * T + slot] = (float)(-0.1 * q + 0.05 * (double)sn);
        if (done_roll) done_roll[a * T + slot] = done ? 1.0f : 0.0f;
    }
    state[i] = done ? init_state[i] : sn;
}

// The acting head, the environment step and the next observation's z-filter in one launch -- the
// glue between two policy forwards of a device-resident rollout (sampling head as
// smx_diaggauss_sample_f32, dynamics as synth_env_step_kernel, filter as
// smx_zfilter_forward_sums_f32: same expressions, same results).
__global__ __launch_bounds__(256) void synth_act_env_step_kernel(smx_synth_act_step_t p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int D = p.D, A = p.A, T = p.T, slot = p.slot;
    if (i >= (long)p.n * D) return;
    const long a = i / D;
    const int k = (int)(i - a * D);
    const float nz = p.noise_scale ? p.noise_scale[a] : 1.0f;
    auto action = [&](int j, float& mu, float& sd) {
        mu = p.mean[a * p.ld_mean + j];
        sd = expf(p.log_var[j]);
        if (p.noise_scale) sd = sd * nz;
        float act = p.eps ? p.eps[a * p.ld_eps + j] * sd + mu : mu;
        if (act == act) act = fminf(fmaxf(act, -1.0f), 1.0f);
        return act;
    };
    float mu, sd;
    const float ac = action(k % A, mu, sd);
    const float s = p.state[i];
    const float drift = 0.01f * (float)(((37 * k) % 17) - 8);
    float sn = (0.9f * s + 0.5f * ac) + drift;
    sn = fminf(fmaxf(sn, -10.0f), 10.0f);
    const bool done = (p.t + 1 >= p.episode_len);
    if (p.obs_roll) {
        p.obs_roll[(a * T + slot) * D + k] = s;
        if (slot + 1 < T) p.obs_roll[(a * T + slot + 1) * D + k] = sn;
    }
    if (k < A) {                                      // here k % A == k
        if (p.act_roll) p.act_roll[(a * T + slot) * A + k] = ac;
        if (p.pd_roll) {
            p.pd_roll[(a * T + slot) * 2 * A + k] = mu;
            p.pd_roll[(a * T + slot) * 2 * A + A + k] = sd;
        }
    }
    if (k == 0) {
        double q = 0.0;
        for (int j = 0; j < A; ++j) {
            float m2, s2;
            const float av = action(j, m2, s2);
            q += (double)av * (double)av;
        }
        if (p.rew_roll) p.rew_roll[a * T + slot] = (float)(-0.1 * q + 0.05 * (double)sn);
        if (p.done_roll) p.done_roll[a * T + slot] = done ? 1.0f : 0.0f;
    }
    const float next = done ? p.init_state[i] : sn;
    p.state[i] = next;
    if (p.xn_out) {
        float z = next;
        if (p.zsum) {
            const float c = p.zcount[0];
            const float m = p.zsum[k] / c;
            const float var = p.zsumsq[k] / c - m * m;
            float sz = sqrtf(var);
            if (sz == sz) sz = fmaxf(sz, p.zeps);
            z = zclamp(next, m, sz);
        }
        p.xn_out[i] = z;
    }
}

// The same step with the policy's OUTPUT LAYER folded in: one workgroup per actor forms mean = act(h2 . W3^T + b3)
// itself (A <= 32 outputs, eight lanes per output, a fixed-order tree) and goes on as synth_act_env_step_kernel does --
// an acting step is then three dependent launches (hidden layer 1, hidden layer 2, this) instead of four.
__global__ __launch_bounds__(256) void synth_act_head_step_kernel(smx_synth_act_step_t p, const float* __restrict__ W3,
                                                                  const float* __restrict__ b3,
                                                                  const float* __restrict__ h2, long ld_h2, int H2,
                                                                  int out_act) {
    __shared__ float s_act[32];
    const int tid = threadIdx.x;
    const long a = blockIdx.x;
    const int D = p.D, A = p.A, T = p.T, slot = p.slot;
    {
        const int j = tid >> 3, l = tid & 7;
        float acc = 0.f;
        if (j < A) {
            const float* hr = h2 + a * ld_h2;
            const float* wr = W3 + (size_t)j * H2;
            for (int k = l; k < H2; k += 8) acc = fmaf(hr[k], wr[k], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (j < A && l == 0) {
            float mu = acc + b3[j];
            if (out_act == SMX_ACT_TANH) mu = tanhf(mu);
            else if (out_act == SMX_ACT_RELU) mu = mu < 0.f ? 0.f : mu;
            const float nz = p.noise_scale ? p.noise_scale[a] : 1.0f;
            float sd = expf(p.log_var[j]);
            if (p.noise_scale) sd = sd * nz;
            float act = p.eps ? p.eps[a * p.ld_eps + j] * sd + mu : mu;
            if (act == act) act = fminf(fmaxf(act, -1.0f), 1.0f);
            s_act[j] = act;
            if (p.act_roll) p.act_roll[(a * T + slot) * A + j] = act;
            if (p.pd_roll) {
                p.pd_roll[(a * T + slot) * 2 * A + j] = mu;
                p.pd_roll[(a * T + slot) * 2 * A + A + j] = sd;
            }
        }
    }
    __syncthreads();
    const bool done = (p.t + 1 >= p.episode_len);
    for (int k = tid; k < D; k += 256) {
        const long i = a * D + k;
        const float ac = s_act[k % A];
        const float s = p.state[i];
        const float drift = 0.01f * (float)(((37 * k) % 17) - 8);
        float sn = (0.9f * s + 0.5f * ac) + drift;
        sn = fminf(fmaxf(sn, -10.0f), 10.0f);
        if (p.obs_roll) {
            p.obs_roll[(a * T + slot) * D + k] = s;
            if (slot + 1 < T) p.obs_roll[(a * T + slot + 1) * D + k] = sn;
        }
        if (k == 0) {
            double q = 0.0;
            for (int j = 0; j < A; ++j) q += (double)s_act[j] * (double)s_act[j];
            if (p.rew_roll) p.rew_roll[a * T + slot] = (float)(-0.1 * q + 0.05 * (double)sn);
            if (p.done_roll) p.done_roll[a * T + slot] = done ? 1.0f : 0.0f;
        }
        const float next = done ? p.init_state[i] : sn;
        p.state[i] = next;
        if (p.xn_out) {
            float z = next;
            if (p.zsum) {
                const float c = p.zcount[0];
                const float m = p.zsum[k] / c;
                const float var = p.zsumsq[k] / c - m * m;
                float sz = sqrtf(var);
                if (sz == sz) sz = fmaxf(sz, p.zeps);
                z = zclamp(next, m, sz);
            }
            p.xn_out[i] = z;
        }
    }
}

inline unsigned row_blocks(long n, int width) {
    const long items = n * ((width + ROW_SEG - 1) / ROW_SEG);
    long b = (items + 3) / 4;  // 4 waves (row segments) per block
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

inline int can_vec(const void* a, const void* b, int width) {
    return (width % 4 == 0) && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0);
}

inline unsigned row_blocks_bytes(long n, long row_bytes) {
    const long items = n * ((row_bytes + ROW_SEG_BYTES - 1) / ROW_SEG_BYTES);
    long b = (items + 3) / 4;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

inline int byte_gran(const void* a, const void* b, long row_bytes) {
    const uintptr_t u = (uintptr_t)a | (uintptr_t)b | (uintptr_t)row_bytes;
    return (u & 15) == 0 ? 16 : ((u & 3) == 0 ? 4 : 1);
}

}  // namespace

extern "C" int smx_ring_insert_f32(float* table, int64_t capacity, int32_t width, int64_t cursor,
                                   const float* src, int64_t n, smx_stream_t stream) {
    SMX_REQUIRE(table && src, SMX_E_NULL);
    SMX_REQUIRE(capacity > 0 && width > 0 && cursor >= 0 && cursor < capacity && n > 0 &&
                    n <= capacity, SMX_E_SHAPE);
    hipLaunchKernelGGL(ring_insert_kernel, dim3(row_blocks(n, width)), dim3(256), 0, smx_s(stream), table,
                       (long)capacity, width, (long)cursor, src, (long)n, can_vec(table, src, width));
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_gather_rows_f32(const float* table, int64_t capacity, int32_t width,
                                   const int64_t* idx, int64_t n, float* dst, smx_stream_t stream) {
    SMX_REQUIRE(table && idx && dst, SMX_E_NULL);
    SMX_REQUIRE(capacity > 0 && width > 0 && n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(row_blocks(n, width)), dim3(256), 0, smx_s(stream), table,
                       (long)capacity, width, idx, (long)n, dst, can_vec(table, dst, width));
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_philox4x32_10(const uint32_t* ctr_key, int64_t n, uint32_t* out, smx_stream_t stream) {
    SMX_REQUIRE(ctr_key && out, SMX_E_NULL);
    SMX_REQUIRE(n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(philox_kat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, smx_s(stream), ctr_key, (long)n, out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_uniform_indices(int64_t* idx, int64_t n, int64_t len, uint64_t seed,
                                   uint64_t offset, smx_stream_t stream) {
    SMX_REQUIRE(idx, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && len > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(uniform_indices_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       smx_s(stream), idx, (long)n, (uint64_t)len, seed, offset);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_uniform_gather_multi(const smx_gather_job_t* jobs, int32_t njobs, int64_t capacity, int64_t rows,
                                        const int64_t* idx, int64_t len, uint64_t seed, uint64_t offset,
                                        int64_t* idx_out, smx_stream_t stream) {
    SMX_REQUIRE(jobs, SMX_E_NULL);
    SMX_REQUIRE(njobs >= 1 && njobs <= 8 && capacity > 0 && rows > 0 && (idx || (len > 0 && len <= capacity)), SMX_E_SHAPE);
    GatherJobs J;
    J.n = njobs;
    int base = 0;
    for (int k = 0; k < njobs; ++k) {
        SMX_REQUIRE(jobs[k].table && jobs[k].dst, SMX_E_NULL);
        SMX_REQUIRE(jobs[k].row_bytes > 0, SMX_E_SHAPE);
        J.table[k] = (const unsigned char*)jobs[k].table;
        J.dst[k] = (unsigned char*)jobs[k].dst;
        J.row_bytes[k] = (long)jobs[k].row_bytes;
        J.gran[k] = byte_gran(jobs[k].table, jobs[k].dst, (long)jobs[k].row_bytes);
        J.block0[k] = base;
        base += (int)row_blocks_bytes((long)rows, (long)jobs[k].row_bytes);
    }
    for (int k = njobs; k <= 8; ++k) J.block0[k] = base;
    hipLaunchKernelGGL(uniform_gather_multi_kernel, dim3((unsigned)base), dim3(256), 0, smx_s(stream), J, (long)capacity,
                       (long)rows, idx, (uint64_t)len, seed, offset, idx_out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_window_emit_f32(const float* src, int32_t actors, int32_t T, int32_t width,
                                   int32_t start, int32_t n_step, int32_t stride, int32_t W,
                                   float* dst, smx_stream_t stream) {
    SMX_REQUIRE(src && dst, SMX_E_NULL);
    SMX_REQUIRE(actors > 0 && T > 0 && width > 0 && n_step > 0 && stride > 0 && W > 0 && start >= 0 &&
                    start + (long)(W - 1) * stride + n_step <= T, SMX_E_SHAPE);
    const long n = (long)actors * W * n_step;
    hipLaunchKernelGGL(window_emit_kernel, dim3(row_blocks(n, width)), dim3(256), 0, smx_s(stream), src,
                       actors, T, width, start, n_step, stride, W, dst, can_vec(src, dst, width));
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ring_insert_bytes(void* table, int64_t capacity, int64_t row_bytes, int64_t cursor,
                                     const void* src, int64_t n, smx_stream_t stream) {
    SMX_REQUIRE(table && src, SMX_E_NULL);
    SMX_REQUIRE(capacity > 0 && row_bytes > 0 && cursor >= 0 && cursor < capacity && n > 0 && n <= capacity,
                SMX_E_SHAPE);
    hipLaunchKernelGGL(ring_insert_bytes_kernel, dim3(row_blocks_bytes(n, row_bytes)), dim3(256), 0, smx_s(stream),
                       (unsigned char*)table, (long)capacity, (long)row_bytes, (long)cursor,
                       (const unsigned char*)src, (long)n, byte_gran(table, src, row_bytes));
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_gather_rows_bytes(const void* table, int64_t capacity, int64_t row_bytes, const int64_t* idx,
                                     int64_t n, void* dst, smx_stream_t stream) {
    SMX_REQUIRE(table && idx && dst, SMX_E_NULL);
    SMX_REQUIRE(capacity > 0 && row_bytes > 0 && n > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3(row_blocks_bytes(n, row_bytes)), dim3(256), 0, smx_s(stream),
                       (const unsigned char*)table, (long)capacity, (long)row_bytes, idx, (long)n,
                       (unsigned char*)dst, byte_gran(table, dst, row_bytes));
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_window_emit_bytes(const void* src, int32_t actors, int32_t T, int64_t row_bytes, int32_t start,
                                     int32_t n_step, int32_t stride, int32_t W, void* dst, smx_stream_t stream) {
    SMX_REQUIRE(src && dst, SMX_E_NULL);
    SMX_REQUIRE(actors > 0 && T > 0 && row_bytes > 0 && n_step > 0 && stride > 0 && W > 0 && start >= 0 &&
                    start + (long)(W - 1) * stride + n_step <= T, SMX_E_SHAPE);
    const long n = (long)actors * W * n_step;
    hipLaunchKernelGGL(window_emit_bytes_kernel, dim3(row_blocks_bytes(n, row_bytes)), dim3(256), 0, smx_s(stream),
                       (const unsigned char*)src, actors, T, (long)row_bytes, start, n_step, stride, W,
                       (unsigned char*)dst, byte_gran(src, dst, row_bytes));
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_frame_stack_u8(const void* frames, int32_t actors, int32_t R, int64_t frame_bytes, int32_t n_stack,
                                  const int32_t* episode_first, int32_t start, int32_t n_step, int32_t stride, int32_t W,
                                  void* dst, smx_stream_t stream) {
    SMX_REQUIRE(frames && dst, SMX_E_NULL);
    SMX_REQUIRE(actors > 0 && R > 0 && frame_bytes > 0 && n_stack > 0 && n_step > 0 && stride > 0 && W > 0 && start >= 0 &&
                    start + (long)(W - 1) * stride + n_step <= R, SMX_E_SHAPE);
    const long n = (long)actors * W * n_step * n_stack;
    hipLaunchKernelGGL(frame_stack_kernel, dim3(row_blocks_bytes(n, frame_bytes)), dim3(256), 0, smx_s(stream),
                       (const unsigned char*)frames, actors, R, (long)frame_bytes, n_stack, (const int*)episode_first, start,
                       n_step, stride, W, (unsigned char*)dst, byte_gran(frames, dst, frame_bytes));
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_synth_frame_u8(const float* s0, int64_t ld_s0, int32_t n, int32_t C, int32_t H, int32_t W, int32_t t,
                                  void* dst, int64_t ld_dst, smx_stream_t stream) {
    SMX_REQUIRE(s0 && dst, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && C > 0 && H > 0 && W > 0 && t >= 0 && ld_s0 > 0 && ld_dst >= (int64_t)C * H * W, SMX_E_SHAPE);
    const long total = (long)n * C * H * W;
    hipLaunchKernelGGL(synth_frame_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, smx_s(stream), s0,
                       (long)ld_s0, n, C, H, W, t, (unsigned char*)dst, (long)ld_dst);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_synth_act_env_step_f32(const smx_synth_act_step_t* args, smx_stream_t stream) {
    SMX_REQUIRE(args, SMX_E_NULL);
    const smx_synth_act_step_t& p = *args;
    SMX_REQUIRE(p.state && p.init_state && p.mean && p.log_var, SMX_E_NULL);
    SMX_REQUIRE(p.n > 0 && p.D > 0 && p.A > 0 && p.A <= p.D && p.T > 0 && p.slot >= 0 && p.slot < p.T &&
                    p.episode_len > 0 && p.ld_mean >= p.A && (!p.eps || p.ld_eps >= p.A), SMX_E_SHAPE);
    SMX_REQUIRE(!p.zsum || (p.zsumsq && p.zcount && p.xn_out), SMX_E_NULL);
    const long total = (long)p.n * p.D;
    hipLaunchKernelGGL(synth_act_env_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       smx_s(stream), p);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_synth_act_env_step_head_f32(const smx_synth_act_step_t* args, const float* W3, const float* b3,
                                               const float* h2, int64_t ld_h2, int32_t H2, int32_t out_act,
                                               smx_stream_t stream) {
    SMX_REQUIRE(args && W3 && b3 && h2, SMX_E_NULL);
    const smx_synth_act_step_t& p = *args;
    SMX_REQUIRE(p.state && p.init_state && p.log_var, SMX_E_NULL);
    SMX_REQUIRE(p.n > 0 && p.D > 0 && p.A > 0 && p.A <= 32 && p.A <= p.D && p.T > 0 && p.slot >= 0 && p.slot < p.T &&
                    p.episode_len > 0 && H2 > 0 && ld_h2 >= H2 && (!p.eps || p.ld_eps >= p.A), SMX_E_SHAPE);
    SMX_REQUIRE(!p.zsum || (p.zsumsq && p.zcount && p.xn_out), SMX_E_NULL);
    hipLaunchKernelGGL(synth_act_head_step_kernel, dim3((unsigned)p.n), dim3(256), 0, smx_s(stream), p, W3, b3, h2,
                       (long)ld_h2, H2, out_act);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_synth_env_step_f32(float* state, const float* init_state, const float* actions,
                                      int32_t n, int32_t D, int32_t A, int32_t t,
                                      int32_t episode_len, int32_t slot, int32_t T, float* obs_roll,
                                      float* act_roll, float* rew_roll, float* done_roll,
                                      smx_stream_t stream) {
    SMX_REQUIRE(state && init_state && actions, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && D > 0 && A > 0 && A <= D && T > 0 && slot >= 0 && slot < T && episode_len > 0,
                SMX_E_SHAPE);
    const long total = (long)n * D;
    hipLaunchKernelGGL(synth_env_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       smx_s(stream), state, init_state, actions, n, D, A, t, episode_len, slot, T,
                       obs_roll, act_roll, rew_roll, done_roll);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
