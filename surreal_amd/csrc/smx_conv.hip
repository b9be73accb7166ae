// CNN stem pieces (surreal/model/model_builders/builders.py:8-33: Conv2d(16, k8, s4)-ReLU-
// Conv2d(32, k4, s2)-ReLU-Flatten-Linear(cnn_feature_dim)-ReLU on the camera image scaled by
// 1/255, ppo_net.py:368-375).  The convolutions run as GEMMs on the FP32-MFMA layer kernel
// (smx_linear_f32 / smx_linear_wgrad_f32); this file holds the data movement around them:
//   * im2col: patches -> rows, reading the uint8 camera frames directly (the reference first
//     converts the whole batch to fp32, ppo.py:436-441) and applying x / 255 on the way;
//   * col2im: the data gradient of a convolution as a GATHER over the (kernel/stride)^2 output
//     positions that saw each input element -- deterministic, no atomics -- with the ReLU mask of
//     the producing layer fused;
//   * the Flatten order: activations are kept channel-last [frame, pixel, channel] (that is what
//     a GEMM over patch rows writes), torch flattens channel-first, so the Linear's weight is
//     re-indexed [out, c*P + p] <-> [out, p*C + c].
#include "smx_common.h"
#include <stdlib.h>

namespace {

struct ConvGeom {
    int C, Hin, Win, kh, kw, stride, Ho, Wo;
};

// SRC_U8: src is uint8 NCHW frames, value = float(u8) / 255.0f; else fp32.
// CHANNEL_LAST: src is [F, Hin*Win, C] (an activation of ours); else [F, C, Hin, Win].
// IDX: unsigned (32-bit index arithmetic) whenever the patch matrix and the source have fewer than 2^31 elements --
// the element -> (row, c, i, j) decomposition is five integer divisions, and 64-bit ones (software sequences of ~60
// instructions each on this ISA) made these pure data-movement kernels ALU-bound at 1.2 TB/s.
template <bool SRC_U8, bool CHANNEL_LAST, typename IDX>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ src, ConvGeom g,
                                                     long long total_, float scale_div,
                                                     float* __restrict__ cols) {
    const IDX K = (IDX)(g.C * g.kh * g.kw), total = (IDX)total_;
    const IDX P = (IDX)(g.Ho * g.Wo);
    for (IDX idx = (IDX)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (IDX)gridDim.x * 256) {
        const IDX row = idx / K;
        const int k = (int)(idx - row * K);
        const int c = k / (g.kh * g.kw), ij = k - c * (g.kh * g.kw);
        const int i = ij / g.kw, j = ij - i * g.kw;
        const IDX f = row / P;
        const int p = (int)(row - f * P);
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        const int y = oy * g.stride + i, x = ox * g.stride + j;
        const IDX s = CHANNEL_LAST
                          ? ((f * g.Hin + y) * g.Win + x) * g.C + c
                          : ((f * g.C + c) * g.Hin + y) * (IDX)g.Win + x;
        float v = SRC_U8 ? (float)static_cast<const unsigned char*>(src)[s]
                         : static_cast<const float*>(src)[s];
        if (scale_div != 0.f) v = v / scale_div;
        cols[idx] = v;
    }
}

// dX[f, y, x, c] = mask * sum over (i, j) with oy*stride + i == y, ox*stride + j == x of
//                  dcols[(f, oy, ox), c*kh*kw + i*kw + j]          (channel-last dX)
template <typename IDX>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, ConvGeom g,
                                                     long long total_,
                                                     const float* __restrict__ relu_of,
                                                     float* __restrict__ dx) {
    const IDX K = (IDX)(g.C * g.kh * g.kw), total = (IDX)total_;
    for (IDX idx = (IDX)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (IDX)gridDim.x * 256) {
        const int c = (int)(idx % (IDX)g.C);
        const IDX pix = idx / (IDX)g.C;
        const IDX f = pix / (IDX)(g.Hin * g.Win);
        const int yx = (int)(pix - f * (IDX)(g.Hin * g.Win));
        const int y = yx / g.Win, x = yx - y * g.Win;
        float acc = 0.f;
        for (int i = y % g.stride; i < g.kh; i += g.stride) {
            const int oy = (y - i) / g.stride;
            if (y - i < 0 || oy >= g.Ho) continue;
            for (int j = x % g.stride; j < g.kw; j += g.stride) {
                const int ox = (x - j) / g.stride;
                if (x - j < 0 || ox >= g.Wo) continue;
                acc += dcols[((f * g.Ho + oy) * g.Wo + ox) * K + (IDX)(c * (g.kh * g.kw) + i * g.kw + j)];
            }
        }
        if (relu_of) acc = (relu_of[idx] > 0.f) ? acc : 0.f;
        dx[idx] = acc;
    }
}

// to_channel_last != 0: out[o, p*C + c] = in[o, c*P + p] ; else the inverse
__global__ __launch_bounds__(256) void flatten_order_kernel(const float* __restrict__ in, int O,
                                                            int C, int P, int to_channel_last,
                                                            float* __restrict__ out) {
    const long long total = (long long)O * C * P;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const long long o = idx / (C * P);
        const int r = (int)(idx - o * (C * P));
        if (to_channel_last) {
            const int p = r / C, c = r - p * C;
            out[idx] = in[o * (C * P) + c * P + p];
        } else {
            const int c = r / P, p = r - c * P;
            out[idx] = in[o * (C * P) + p * C + c];
        }
    }
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

// ---------------------------------------------------------------------------------------------
// The first convolution over uint8 camera frames as an IMPLICIT GEMM: y[(f, oy, ox), o] =
// relu(b[o] + sum_k W[o][k] * frame[f][c][oy s + i][ox s + j] / 255), k = (c, i, j).  The materialised
// patch matrix of that layer is 2.2 GB at 7168 frames (1.7 ms to write, and the GEMM over it reads it
// back at the HBM rate); here a lane gathers its share of a patch straight from the frame.
//   * v_mfma_f32_16x16x4: M = 16 consecutive patches (A operand), N = the <= 16 output channels (B);
//   * lane (i = lane & 15, kq = lane >> 4) fetches, per group g of 16 k's, ONE dword = the four bytes
//     k = 16 g + 4 kq + s, s = 0..3 (the same (c, row) of the patch: kw % 4 == 0; aligned: W, stride % 4
//     == 0) -- the k's of four consecutive MFMA steps; its weight registers hold W[lane & 15][the same k's];
//   * u8 / 255.0f exactly: q = x r, q' = fma(fma(-q, 255, x), r, q) with r = fl(1 / 255) is the correctly
//     rounded quotient for every byte (all 256 checked), a multiply alone is wrong for 126 of them;
//   * the weights stay in registers: a wavefront walks many 16-patch tiles (persistent), the next tile's
//     dwords are requested before the current tile's MFMAs.
// ---------------------------------------------------------------------------------------------
#define MFMA16C(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// q = n / d, r = n % d; 32-bit when the caller knows n < 2^31 (`small`, launch-uniform): a 64-bit division is a
// ~60-instruction software sequence and the implicit-GEMM kernels decompose a row index per lane and tile
__device__ __forceinline__ void divmod_idx(long long n, int d, bool small, long long& q, int& r) {
    if (small) {
        const unsigned uq = (unsigned)n / (unsigned)d;
        q = uq;
        r = (int)((unsigned)n - uq * (unsigned)d);
    } else {
        q = n / d;
        r = (int)(n - q * d);
    }
}

__device__ __forceinline__ float u8_div255(unsigned x, int byte) {
    const float v = (float)((x >> (8 * byte)) & 0xffu);
    const float r = 1.0f / 255.0f;                      // compile-time constant, correctly rounded
    const float q = v * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, v), r, q);
}

template <int NG>      // K = 16 NG
__global__ __launch_bounds__(256) void conv_u8_fwd_kernel(const unsigned char* __restrict__ frames, ConvGeom g,
                                                          long long rows, const float* __restrict__ W,
                                                          const float* __restrict__ bias, int cout,
                                                          float* __restrict__ y, const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int K = 16 * NG;
    // weights of output channel i (zero past cout) and the byte offset of each group's dword inside a frame
    float4 w[NG];
    int koff[NG];
#pragma unroll
    for (int gidx = 0; gidx < NG; ++gidx) {
        const int k = 16 * gidx + 4 * kq;
        w[gidx] = (i < cout) ? *reinterpret_cast<const float4*>(W + (size_t)i * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = k / (g.kh * g.kw), ij = k - c * (g.kh * g.kw);
        const int ki = ij / g.kw, kj = ij - ki * g.kw;
        koff[gidx] = (c * g.Hin + ki) * g.Win + kj;
    }
    const float bv = (i < cout) ? bias[i] : 0.f;
    const int P = g.Ho * g.Wo;
    const long long ntiles = (rows + 15) >> 4;
    auto base_of = [&](long long tile) -> long long {
        long long row = tile * 16 + i;
        if (row >= rows) row = rows - 1;                 // clamped: loaded, never stored
        long long f;
        int p;
        divmod_idx(row, P, rows < (1ll << 31), f, p);
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        return (f * g.C * g.Hin + (long long)oy * g.stride) * g.Win + ox * g.stride;
    };
    unsigned cur[NG], nxt[NG];
    long long tile = wave;
    if (tile < ntiles) {
        const unsigned char* b = frames + base_of(tile);
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx) cur[gidx] = *reinterpret_cast<const unsigned*>(b + koff[gidx]);
    }
    for (; tile < ntiles; tile += nwaves) {
        const long long tn = tile + nwaves;
        {
            const unsigned char* b = frames + base_of(tn < ntiles ? tn : tile);
#pragma unroll
            for (int gidx = 0; gidx < NG; ++gidx) nxt[gidx] = *reinterpret_cast<const unsigned*>(b + koff[gidx]);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx) {
            const unsigned x = cur[gidx];
            acc = MFMA16C(u8_div255(x, 0), w[gidx].x, acc);
            acc = MFMA16C(u8_div255(x, 1), w[gidx].y, acc);
            acc = MFMA16C(u8_div255(x, 2), w[gidx].z, acc);
            acc = MFMA16C(u8_div255(x, 3), w[gidx].w, acc);
        }
        // C fragment: lane holds patches 4 kq + r (r = 0..3) of output channel i
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long row = tile * 16 + 4 * kq + r;
            float v = acc[r] + bv;
            v = v < 0.f ? 0.f : v;
            if (row < rows && i < cout) y[row * cout + i] = v;
        }
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx) cur[gidx] = nxt[gidx];
    }
}

// ---------------------------------------------------------------------------------------------
// The first convolution's WEIGHT GRADIENT from the uint8 frames (no patch matrix): dW[o][k] = sum over patch rows of
// dy[row][o] * patch[row][k], db[o] = sum of dy[row][o].  v_mfma_f32_16x16x4 with M = output channel, N = 16 of the K
// patch columns, the MFMA k dimension = four patch rows per step:
//   * a wavefront takes 16 patch rows at a time: lane (i = row, kq) fetches the row's bytes as dwords exactly like the
//     forward kernel and drops them RAW into a 16 x K byte tile in LDS; the B fragments are then byte reads,
//     tile[4 kq + s][16 t + i], converted with the exact u8 / 255;
//   * the A fragments are dy[r0 + 4 kq + s][i] (64-byte runs);
//   * NG accumulators (16 x 16 each) stay in registers over all the groups a wavefront walks; the four wavefronts of a
//     workgroup are summed through LDS and a workgroup writes ONE partial [cout][K] (+ [cout] for the bias); a second
//     launch adds the partials in a fixed order (smx_gemm.hip's split-K reduce contract: deterministic).
// ---------------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(256) void conv_u8_wgrad_kernel(const unsigned char* __restrict__ frames, ConvGeom g,
                                                            long long rows, const float* __restrict__ dy, int cout,
                                                            float* __restrict__ part, float* __restrict__ dbpart,
                                                            const int* __restrict__ stop) {
    if (stop && *stop) return;
    constexpr int K = 16 * NG;
    constexpr int RS = K + 16;                          // byte row stride of the LDS tile (bank spread)
    __shared__ unsigned char tile[4][16 * RS];
    __shared__ float comb[4][16 * K];                   // the four wavefronts' accumulators, then their sum
    __shared__ float dbs[4][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long nwaves = (long long)gridDim.x * 4;
    int koff[NG];
#pragma unroll
    for (int gidx = 0; gidx < NG; ++gidx) {
        const int k = 16 * gidx + 4 * kq;
        const int c = k / (g.kh * g.kw), ij = k - c * (g.kh * g.kw);
        const int ki = ij / g.kw, kj = ij - ki * g.kw;
        koff[gidx] = (c * g.Hin + ki) * g.Win + kj;
    }
    const int P = g.Ho * g.Wo;
    const long long ngroups = (rows + 15) >> 4;
    auto base_of = [&](long long grp) -> long long {
        long long row = grp * 16 + i;
        if (row >= rows) row = rows - 1;                // clamped: its dy is taken as zero
        long long f;
        int p;
        divmod_idx(row, P, rows < (1ll << 31), f, p);
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        return (f * g.C * g.Hin + (long long)oy * g.stride) * g.Win + ox * g.stride;
    };
    f32x4 acc[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.f;
    unsigned char* mytile = tile[wv];
    unsigned cur[NG], nxt[NG];
    long long grp = wave;
    // (unconditional loads from clamped addresses, zeroed by a select: a load under a lane mask is waited for on the spot)
#define DY_REQUEST(G, DST, S4)                                                        \
    do {                                                                              \
        const long long row_ = (G) * 16 + 4 * kq + (S4);                              \
        const bool ok_ = row_ < rows && i < cout;                                     \
        const float v_ = dy[(ok_ ? row_ : 0) * cout + (i < cout ? i : 0)];            \
        DST = ok_ ? v_ : 0.f;                                                         \
    } while (0)
    float na0 = 0.f, na1 = 0.f, na2 = 0.f, na3 = 0.f;
    if (grp < ngroups) {
        const unsigned char* b = frames + base_of(grp);
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx) cur[gidx] = *reinterpret_cast<const unsigned*>(b + koff[gidx]);
        DY_REQUEST(grp, na0, 0); DY_REQUEST(grp, na1, 1); DY_REQUEST(grp, na2, 2); DY_REQUEST(grp, na3, 3);
    }
    for (; grp < ngroups; grp += nwaves) {
        const long long gn = grp + nwaves;
        {
            const unsigned char* b = frames + base_of(gn < ngroups ? gn : grp);
#pragma unroll
            for (int gidx = 0; gidx < NG; ++gidx) nxt[gidx] = *reinterpret_cast<const unsigned*>(b + koff[gidx]);
        }
        // dy of the group's rows 4 kq + s, channel i: requested ONE GROUP AHEAD like the frame bytes (the wavefront has
        // its SIMD to itself -- one partial per wavefront bounds the grid -- so an exposed L2 / HBM round trip per group
        // was half of the group's time).  Named scalars: a loop-carried array is left in scratch memory by hipcc.
        float a[4] = {na0, na1, na2, na3};
        dbacc += a[0]; dbacc += a[1]; dbacc += a[2]; dbacc += a[3];
        {
            const long long g2 = gn < ngroups ? gn : grp;
            DY_REQUEST(g2, na0, 0); DY_REQUEST(g2, na1, 1); DY_REQUEST(g2, na2, 2); DY_REQUEST(g2, na3, 3);
        }
        // raw bytes of patch row i, columns 16 g + 4 kq .. + 3  -> the wavefront's LDS tile
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx)
            *reinterpret_cast<unsigned*>(mytile + i * RS + 16 * gidx + 4 * kq) = cur[gidx];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // LDS is in order per wavefront; the tile is private
#pragma unroll
        for (int t = 0; t < NG; ++t) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const unsigned bt = mytile[(4 * kq + s4) * RS + 16 * t + i];
                acc[t] = MFMA16C(a[s4], u8_div255(bt, 0), acc[t]);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx) cur[gidx] = nxt[gidx];
    }
#undef DY_REQUEST
    // lane (i, kq) holds dW[o = 4 kq + r][k = 16 t + i]; the bias partial: sum over the four kq lane groups
    dbacc += __shfl_xor(dbacc, 16, 64);
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (kq == 0) dbs[wv][i] = dbacc;
#pragma unroll
    for (int t = 0; t < NG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) comb[wv][(4 * kq + r) * K + 16 * t + i] = acc[t][r];
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * cout * K;
    for (int idx = threadIdx.x; idx < cout * K; idx += 256)
        out[idx] = ((comb[0][idx] + comb[1][idx]) + comb[2][idx]) + comb[3][idx];
    if (threadIdx.x < cout)
        dbpart[(size_t)blockIdx.x * cout + threadIdx.x] =
            ((dbs[0][threadIdx.x] + dbs[1][threadIdx.x]) + dbs[2][threadIdx.x]) + dbs[3][threadIdx.x];
}

// The split partials of a weight gradient summed in a fixed order on a 2-D grid: a workgroup owns RED_EL consecutive
// elements, its 256 threads are RED_EL elements x RED_SL slices of the partial index; a thread sums its slice on four
// interleaved chains, the slices are combined through LDS in slice order.  (One thread per element walking all 512 - 1024
// partials was 13 - 33 workgroups on 256 CUs: 41 / 66 us per launch, 2 ms per learn at configs[3].)
constexpr int RED_EL = 16, RED_SL = 256 / RED_EL;
__device__ __forceinline__ float reduce_slices(const float* __restrict__ src, size_t stride, int splits, bool valid) {
    __shared__ float red[RED_SL][RED_EL];
    const int el = threadIdx.x % RED_EL, sl = threadIdx.x / RED_EL;
    const int per = (splits + RED_SL - 1) / RED_SL;
    const int s0 = sl * per, s1 = s0 + per < splits ? s0 + per : splits;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (valid) {
        int sidx = s0;
        for (; sidx + 3 < s1; sidx += 4) {
            v0 += src[(size_t)sidx * stride];
            v1 += src[(size_t)(sidx + 1) * stride];
            v2 += src[(size_t)(sidx + 2) * stride];
            v3 += src[(size_t)(sidx + 3) * stride];
        }
        for (; sidx < s1; ++sidx) v0 += src[(size_t)sidx * stride];
    }
    red[sl][el] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    float v = 0.f;
    if (sl == 0) {
        v = red[0][el];
#pragma unroll
        for (int j = 1; j < RED_SL; ++j) v += red[j][el];
    }
    return v;
}

// out[e] = sum over s of part[s][e], fixed order
__global__ __launch_bounds__(256) void conv_partial_reduce_kernel(const float* __restrict__ part,
                                                                  const float* __restrict__ dbpart, int splits,
                                                                  int nW, int nB, float* __restrict__ dW,
                                                                  float* __restrict__ db,
                                                                  const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int e = blockIdx.x * RED_EL + threadIdx.x % RED_EL;
    const bool valid = e < nW + nB;
    const float* src = e < nW ? part + e : dbpart + (e - nW);
    const int stride = e < nW ? nW : nB;
    const float v = reduce_slices(src, (size_t)stride, splits, valid);
    if (threadIdx.x >= RED_EL || !valid) return;
    if (e < nW) dW[e] = v;
    else if (db) db[e - nW] = v;
}

constexpr int CONV_WGRAD_BLOCKS = 512;      // two workgroups per CU (a wavefront's byte unpacking hides under its SIMD partner's MFMAs): 512 partials

// ---------------------------------------------------------------------------------------------
// The SECOND convolution (fp32 channel-last source [F, Hin*Win, 16], e.g. the first one's output) as implicit GEMMs.
// A patch is kh*kw positions of 16 contiguous channels: lane (i = patch, kq) fetches, per position p, the float4 of
// channels 4 kq .. 4 kq + 3 -- the k's of four consecutive MFMA steps -- so the column order inside the kernels is
// (p, c) while the weight stays in torch's [o][c][p] order (each lane picks its W[o][4 kq + s][p] once, into registers).
// No patch matrix (594 MB at 7168 frames), no im2col launch (0.5 ms), and the GEMM no longer streams that matrix.
// ---------------------------------------------------------------------------------------------
template <int NP, int NT>      // NP = kh*kw positions (<= 16), NT = 16-channel output tiles (cout <= 16 NT)
__global__ __launch_bounds__(256) void conv_cl_fwd_kernel(const float* __restrict__ src, ConvGeom g, long long rows,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          int cout, float* __restrict__ y, const int* __restrict__ stop) {
    if (stop && *stop) return;
    // the weight fragments in LDS, fragment order [nt][p][lane] (one ds_read_b128 per four MFMAs; in registers they are
    // 128 VGPRs at cout = 32 and left two wavefronts per SIMD to hide the patch gather's latency), and the next tile's
    // patch words requested one tile ahead
    __shared__ float4 ws[NT * NP * 64];
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (int e = threadIdx.x; e < NT * NP * 64; e += 256) {
        const int l = e & 63, p = (e >> 6) % NP, nt = (e >> 6) / NP;
        const int o = 16 * nt + (l & 15);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < cout) {
            const float* q = W + ((size_t)o * 16 + 4 * (l >> 4)) * NP + p;       // W[o][c][p], c = 4 kq + s
            v = make_float4(q[0], q[NP], q[2 * NP], q[3 * NP]);
        }
        ws[e] = v;
    }
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = 16 * nt + i < cout ? bias[16 * nt + i] : 0.f;
    __syncthreads();
    int poff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) poff[p] = ((p / g.kw) * g.Win + (p % g.kw)) * 16 + 4 * kq;
    const int P = g.Ho * g.Wo;
    const long long ntiles = (rows + 15) >> 4;
    // (named scalars: a loop-carried ARRAY of prefetched words is left in scratch memory by hipcc)
#define PA_ALL(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define PA_DECL(k) float4 pa##k = make_float4(0.f, 0.f, 0.f, 0.f);
#define PA_LOAD(k) if (k < NP) pa##k = *reinterpret_cast<const float4*>(b_ + poff[k < NP ? k : 0]);
#define PA_TAKE(k) if (k < NP) a[k < NP ? k : 0] = pa##k;
#define PA_REQUEST(T)                                                                                       \
    do {                                                                                                    \
        long long row_ = (T) * 16 + i;                                                                      \
        if (row_ >= rows) row_ = rows - 1;                                                                  \
        long long f_;                                                                                       \
        int pp_;                                                                                            \
        divmod_idx(row_, P, rows < (1ll << 31), f_, pp_);                                                   \
        const int oy_ = pp_ / g.Wo, ox_ = pp_ - oy_ * g.Wo;                                                 \
        const float* b_ = src + ((f_ * g.Hin + (long long)oy_ * g.stride) * g.Win + ox_ * g.stride) * 16;  \
        PA_ALL(PA_LOAD)                                                                                     \
    } while (0)
    PA_ALL(PA_DECL)
    long long tile = wave;
    if (tile < ntiles) PA_REQUEST(tile);
    for (; tile < ntiles; tile += nwaves) {
        float4 a[NP];
        PA_ALL(PA_TAKE)
        {
            const long long tn = tile + nwaves < ntiles ? tile + nwaves : tile;
            PA_REQUEST(tn);
        }
        // (always 0, but not to the compiler: the weight reads must stay LDS reads inside the loop)
        const int z = __builtin_amdgcn_readfirstlane((int)((unsigned long long)tile >> 62));
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 wv = ws[(nt * NP + p) * 64 + lane + z];
                acc[nt] = MFMA16C(a[p].x, wv.x, acc[nt]);
                acc[nt] = MFMA16C(a[p].y, wv.y, acc[nt]);
                acc[nt] = MFMA16C(a[p].z, wv.z, acc[nt]);
                acc[nt] = MFMA16C(a[p].w, wv.w, acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int o = 16 * nt + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long orow = tile * 16 + 4 * kq + r;
                float v = acc[nt][r] + bv[nt];
                v = v < 0.f ? 0.f : v;
                if (orow < rows && o < cout) y[orow * cout + o] = v;
            }
        }
    }
#undef PA_ALL
#undef PA_DECL
#undef PA_LOAD
#undef PA_TAKE
#undef PA_REQUEST
}

// weight gradient of the same layer: per-WAVEFRONT partials dW[o][(p, c)] (+ db), summed and re-ordered to torch's
// [o][c][p] by conv_cl_wgrad_reduce_kernel.  The 16 x (16 NP) float tile of a group's patches goes through LDS (the
// forward kernel's float4 gathers in, the B fragments' single floats out), as in conv_u8_wgrad_kernel.
template <int NP, int NT>
__global__ __launch_bounds__(256) void conv_cl_wgrad_kernel(const float* __restrict__ src, ConvGeom g, long long rows,
                                                            const float* __restrict__ dy, int cout,
                                                            float* __restrict__ part, float* __restrict__ dbpart,
                                                            const int* __restrict__ stop) {
    if (stop && *stop) return;
    constexpr int K = 16 * NP;
    constexpr int RS = K + 4;                            // float row stride of the LDS tile
    extern __shared__ float cl_tile[];                   // [4 wavefronts][16][RS]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long nwaves = (long long)gridDim.x * 4;
    float* mytile = cl_tile + (size_t)wv * 16 * RS;
    int poff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) poff[p] = ((p / g.kw) * g.Win + (p % g.kw)) * 16 + 4 * kq;
    const int P = g.Ho * g.Wo;
    const long long ngroups = (rows + 15) >> 4;
    f32x4 acc[NT][NP];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[nt][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dbacc[nt] = 0.f;
    // a group's operands are requested one group AHEAD (a wavefront has its SIMD to itself here -- 128 accumulator
    // registers -- so nothing else hides the gather's latency): the patch words and the dy words of group g + 1 are in
    // flight while group g's 128 MFMAs issue
    // (written out twice, unconditionally: behind a lambda or a branch hipcc leaves the arrays in scratch memory)
    // (named scalars: a loop-carried ARRAY of prefetched words is left in scratch memory by hipcc)
#define PW_ALL(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define PW_DECL(k) float4 pw##k = make_float4(0.f, 0.f, 0.f, 0.f);
#define PW_LOAD(k) if (k < NP) pw##k = *reinterpret_cast<const float4*>(b + poff[k < NP ? k : 0]);
#define PW_STORE(k) if (k < NP) *reinterpret_cast<float4*>(mytile + i * RS + 16 * k + 4 * kq) = pw##k;
    PW_ALL(PW_DECL)
    float a[NT][4];
    long long grp = wave;
    {
        long long row = (grp < ngroups ? grp : 0) * 16 + i;
        if (row >= rows) row = rows - 1;
        long long f;
        int pp;
        divmod_idx(row, P, rows < (1ll << 31), f, pp);
        const int oy = pp / g.Wo, ox = pp - oy * g.Wo;
        const float* b = src + ((f * g.Hin + (long long)oy * g.stride) * g.Win + ox * g.stride) * 16;
        PW_ALL(PW_LOAD)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const long long r = grp * 16 + 4 * kq + s4;
                const int o = 16 * nt + i;
                const bool ok = r < rows && o < cout;
                const float v = dy[(ok ? r : 0) * cout + (o < cout ? o : 0)];      // unconditional, clamped
                a[nt][s4] = ok ? v : 0.f;
            }
    }
    for (; grp < ngroups; grp += nwaves) {
        PW_ALL(PW_STORE)
        float ac[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                ac[nt][s4] = a[nt][s4];
                dbacc[nt] += ac[nt][s4];
            }
        {
            // the next group (the last iteration re-requests its own: cache hits, results unused)
            const long long gn = grp + nwaves < ngroups ? grp + nwaves : grp;
            long long row = gn * 16 + i;
            if (row >= rows) row = rows - 1;             // clamped: its dy is taken as zero
            long long f;
            int pp;
            divmod_idx(row, P, rows < (1ll << 31), f, pp);
            const int oy = pp / g.Wo, ox = pp - oy * g.Wo;
            const float* b = src + ((f * g.Hin + (long long)oy * g.stride) * g.Win + ox * g.stride) * 16;
            PW_ALL(PW_LOAD)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const long long r = gn * 16 + 4 * kq + s4;
                    const int o = 16 * nt + i;
                    const bool ok = r < rows && o < cout;
                    const float v = dy[(ok ? r : 0) * cout + (o < cout ? o : 0)];
                    a[nt][s4] = ok ? v : 0.f;
                }
        }
        // (LDS operations of one wavefront execute in order; a fence without a memory-clobbering asm statement: an
        // array that is live across one is left in scratch memory by hipcc)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float bval = mytile[(4 * kq + s4) * RS + 16 * p + i];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][p] = MFMA16C(ac[nt][s4], bval, acc[nt][p]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef PW_ALL
#undef PW_DECL
#undef PW_LOAD
#undef PW_STORE
    // lane (i, kq): dW[o = 16 nt + 4 kq + r][(p, c = i)]
    float* out = part + (size_t)wave * (16 * NT) * K;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float d = dbacc[nt];
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        if (kq == 0) dbpart[(size_t)wave * (16 * NT) + 16 * nt + i] = d;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(16 * nt + 4 * kq + r) * K + 16 * p + i] = acc[nt][p][r];
    }
}

// dW[o][c][p] = sum over the wavefront partials of part[s][o][(p, c)]; db likewise.  Fixed order (reduce_slices).
__global__ __launch_bounds__(256) void conv_cl_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                   const float* __restrict__ dbpart, int splits,
                                                                   int OP /* 16 NT */, int NP, int cout,
                                                                   float* __restrict__ dW, float* __restrict__ db,
                                                                   const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int K = 16 * NP;
    const int nW = OP * K;
    const int e = blockIdx.x * RED_EL + threadIdx.x % RED_EL;
    const bool valid = e < nW + OP;
    const float* srcp = e < nW ? part + e : dbpart + (e - nW);
    const int stride = e < nW ? nW : OP;
    const float v = reduce_slices(srcp, (size_t)stride, splits, valid);
    if (threadIdx.x >= RED_EL || !valid) return;
    if (e < nW) {
        const int o = e / K, pc = e - o * K;
        const int p = pc >> 4, c = pc & 15;
        if (o < cout) dW[((size_t)o * 16 + c) * NP + p] = v;
    } else if (db && e - nW < cout) {
        db[e - nW] = v;
    }
}

constexpr int CONV_CL_WGRAD_BLOCKS = 256;   // x 4 wavefronts = 1024 partials

// ---------------------------------------------------------------------------------------------
// The second convolution's DATA GRADIENT as an implicit GEMM (k == 2 stride: every input pixel is seen by exactly
// 2 x 2 kernel positions): dx[(f, y, x), c] = relu'(x_act) * sum over u, v in {0, 1} and o of
//   dy[(f, (y - ki) / s, (x - kj) / s), o] * W[o][c][ki][kj],   ki = y % s + u s,  kj = x % s + v s
// (terms whose output position falls outside the map are absent).  M = 16 input pixels of ONE parity class
// (y % s, x % s) -- they share the four kernel positions, hence the B operand --, N = the 16 input channels,
// K = 4 positions x cout.  The materialised route wrote dcols = dy . W (594 MB at 7168 frames) and gathered it back.
// ---------------------------------------------------------------------------------------------
template <int NH, bool VEC4>          // cout <= 16 NH; VEC4: cout % 4 == 0
__global__ __launch_bounds__(256) void conv_cl_dgrad_kernel(const float* __restrict__ dy, ConvGeom g, long long F,
                                                            const float* __restrict__ W, int cout,
                                                            const float* __restrict__ relu_of, float* __restrict__ dx,
                                                            const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int st = g.stride, NPK = g.kh * g.kw;
    const int na = (g.Hin + st - 1) / st, nb = (g.Win + st - 1) / st;       // class pixels per frame: na x nb (max)
    for (int cls = 0; cls < st * st; ++cls) {
        const int py = cls / st, px = cls - py * st;
        const int ca = (g.Hin - py + st - 1) / st, cb = (g.Win - px + st - 1) / st;   // rows / columns of this class
        if (ca <= 0 || cb <= 0) continue;
        // B operand: lane (c = i, kq) holds W[o = 16 h + 4 kq + s][c][ki][kj] for the class's four positions
        float4 w[4][NH];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ki = py + (q >> 1) * st, kj = px + (q & 1) * st;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float v[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int o = 16 * h + 4 * kq + s4;
                    v[s4] = (o < cout && ki < g.kh && kj < g.kw) ? W[((size_t)o * 16 + i) * NPK + ki * g.kw + kj] : 0.f;
                }
                w[q][h] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        const long long per = (long long)ca * cb;
        const long long npix = F * per;
        const bool small = npix < (1ll << 31);
        const long long ntiles = (npix + 15) >> 4;
        for (long long tile = wave; tile < ntiles; tile += nwaves) {
            long long n = tile * 16 + i;
            const bool inr = n < npix;
            if (!inr) n = npix - 1;
            long long f;
            int ab;
            divmod_idx(n, (int)per, small, f, ab);
            const int a = ab / cb, b = ab - a * cb;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oy = a - (q >> 1), ox = b - (q & 1);
                const bool ok = inr && oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo;
                const long long prow = (f * g.Ho + (ok ? oy : 0)) * g.Wo + (ok ? ox : 0);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const int o0 = 16 * h + 4 * kq;
                    // unconditional loads from clamped addresses, zeroed by selects (a load under a lane mask is
                    // waited for on the spot: eight serial L2 round trips per tile)
                    float4 av;
                    if (VEC4) {                          // cout % 4 == 0: rows are 16-byte aligned
                        av = *reinterpret_cast<const float4*>(dy + prow * cout + (o0 + 3 < cout ? o0 : 0));
                        if (!(ok && o0 + 3 < cout)) av = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        const float* q4 = dy + prow * cout;
                        const float x0 = q4[o0 < cout ? o0 : 0], x1 = q4[o0 + 1 < cout ? o0 + 1 : 0];
                        const float x2 = q4[o0 + 2 < cout ? o0 + 2 : 0], x3 = q4[o0 + 3 < cout ? o0 + 3 : 0];
                        av = make_float4((ok && o0 < cout) ? x0 : 0.f, (ok && o0 + 1 < cout) ? x1 : 0.f,
                                         (ok && o0 + 2 < cout) ? x2 : 0.f, (ok && o0 + 3 < cout) ? x3 : 0.f);
                    }
                    acc = MFMA16C(av.x, w[q][h].x, acc);
                    acc = MFMA16C(av.y, w[q][h].y, acc);
                    acc = MFMA16C(av.z, w[q][h].z, acc);
                    acc = MFMA16C(av.w, w[q][h].w, acc);
                }
            }
            // C fragment: lane (c = i, kq) holds the class pixels 16 tile + 4 kq + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long m = tile * 16 + 4 * kq + r;
                if (m < npix) {
                    long long f2;
                    int ab2;
                    divmod_idx(m, (int)per, small, f2, ab2);
                    const int a2 = ab2 / cb, b2 = ab2 - a2 * cb;
                    const long long pix = (f2 * g.Hin + (long long)(a2 * st + py)) * g.Win + (b2 * st + px);
                    float v = acc[r];
                    if (relu_of) v = (relu_of[pix * 16 + i] > 0.f) ? v : 0.f;      // (uniform branch)
                    dx[pix * 16 + i] = v;
                }
            }
        }
    }
    (void)na; (void)nb;
}

// The same data gradient tiled over CELLS instead of parity classes (stride 2, kernel 4, even Hin / Win -- the
// reference's second convolution): a cell (a, b) is the 2 x 2 block of input pixels (2a + py, 2b + px), and all four of
// them read the SAME four output positions (a - u, b - v), u, v in {0, 1} -- only the kernel taps differ.  A wavefront
// therefore gathers the dy rows of 16 cells ONCE (8 x 16-byte loads per lane) and runs all four classes on them
// (4 x 32 MFMAs against four register-resident weight sets) where the class-major kernel above gathered them four
// times, once per pass over the whole tensor; and the two classes that share a 128-byte line of dx / of the ReLU mask
// (px = 0, 1) now write / read it back to back instead of a whole tensor pass apart.  The next tile's gather is in
// flight under the current tile's MFMAs.  Same sums in the same order per output element: bit-identical results.
template <int NH, bool VEC4>
__global__ __launch_bounds__(256, 3) void conv_cl_dgrad_cells_kernel(const float* __restrict__ dy, ConvGeom g, long long F,
                                                                  const float* __restrict__ W, int cout,
                                                                  const float* __restrict__ relu_of,
                                                                  float* __restrict__ dx, const int* __restrict__ stop) {
    if (stop && *stop) return;
    // B operands of the four classes in LDS, fragment order (one ds_read_b128 per four MFMAs): [cls][q][h][lane].
    // (In registers they cost 128 VGPRs and left two wavefronts per SIMD to hide the HBM latency of the mask reads.)
    __shared__ float4 ws[4 * 4 * NH * 64];
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int NPK = g.kh * g.kw;
    const int ca = g.Hin >> 1, cb = g.Win >> 1;
    for (int e = threadIdx.x; e < 4 * 4 * NH * 64; e += 256) {
        // entry (cls, q, h, l): lane l = (c = l & 15, kq = l >> 4) holds W[o = 16 h + 4 kq + s][c][ki][kj]
        const int l = e & 63, h = (e >> 6) % NH, q = ((e >> 6) / NH) & 3, cls = (e >> 6) / NH / 4;
        const int ki = (cls >> 1) + (q >> 1) * 2, kj = (cls & 1) + (q & 1) * 2;
        float v[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int o = 16 * h + 4 * (l >> 4) + s4;
            v[s4] = o < cout ? W[((size_t)o * 16 + (l & 15)) * NPK + ki * g.kw + kj] : 0.f;
        }
        ws[e] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    const long long per = (long long)ca * cb;
    const long long ncell = F * per;
    const bool small = ncell < (1ll << 31);
    const long long ntiles = (ncell + 15) >> 4;
    // what a tile reads, requested one tile ahead: its A operands -- lane (cell = i, kq) holds dy[(f, a - u, b - v),
    // o = 16 h + 4 kq + 0..3] (zero outside the map) -- and, for the C fragment's cells (16 tile + 4 kq + r), the
    // 2 x 2 pixels' ReLU-mask values and addresses
    struct Tile {
        float4 av[4][NH];
        float mk[4][4];          // [cls][r]
        long long pix0[4];       // pixel (py, px) = (0, 0) of cell r; < 0: past the end
    };
    auto fetch = [&](long long tile, Tile& T) {
        long long n = tile * 16 + i;
        const bool inr = n < ncell;
        if (!inr) n = ncell - 1;
        long long f;
        int ab;
        divmod_idx(n, (int)per, small, f, ab);
        const int a = ab / cb, b = ab - a * cb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = a - (q >> 1), ox = b - (q & 1);
            const bool ok = inr && oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo;
            const long long prow = (f * g.Ho + (ok ? oy : 0)) * g.Wo + (ok ? ox : 0);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int o0 = 16 * h + 4 * kq;
                float4 v;
                if (VEC4) {
                    v = *reinterpret_cast<const float4*>(dy + prow * cout + (o0 + 3 < cout ? o0 : 0));
                    if (!(ok && o0 + 3 < cout)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    const float* q4 = dy + prow * cout;
                    const float x0 = q4[o0 < cout ? o0 : 0], x1 = q4[o0 + 1 < cout ? o0 + 1 : 0];
                    const float x2 = q4[o0 + 2 < cout ? o0 + 2 : 0], x3 = q4[o0 + 3 < cout ? o0 + 3 : 0];
                    v = make_float4((ok && o0 < cout) ? x0 : 0.f, (ok && o0 + 1 < cout) ? x1 : 0.f,
                                    (ok && o0 + 2 < cout) ? x2 : 0.f, (ok && o0 + 3 < cout) ? x3 : 0.f);
                }
                T.av[q][h] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long m = tile * 16 + 4 * kq + r;
            const bool okm = m < ncell;
            long long f2;
            int ab2;
            divmod_idx(okm ? m : 0, (int)per, small, f2, ab2);
            const int a2 = ab2 / cb, b2 = ab2 - a2 * cb;
            const long long p0 = (f2 * g.Hin + 2ll * a2) * g.Win + 2 * b2;
            T.pix0[r] = okm ? p0 : -1;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls)       // (unconditional loads from valid addresses: cell 0 when past the end)
                T.mk[cls][r] = relu_of ? relu_of[(p0 + (long long)(cls >> 1) * g.Win + (cls & 1)) * 16 + i] : 1.f;
        }
    };
    Tile cur, nxt;
    long long tile = wave;
    if (tile < ntiles) fetch(tile, cur);
    for (; tile < ntiles; tile += nwaves) {
        const long long tn = tile + nwaves;
        if (tn < ntiles) fetch(tn, nxt);
        // (always 0, but not to the compiler: the weight reads below must stay LDS reads inside the loop -- hoisted out
        // of it they would be 128 registers again)
        const int z = __builtin_amdgcn_readfirstlane((int)((unsigned long long)tile >> 62));
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float4 wv = ws[((cls * 4 + q) * NH + h) * 64 + lane + z];
                    acc = MFMA16C(cur.av[q][h].x, wv.x, acc);
                    acc = MFMA16C(cur.av[q][h].y, wv.y, acc);
                    acc = MFMA16C(cur.av[q][h].z, wv.z, acc);
                    acc = MFMA16C(cur.av[q][h].w, wv.w, acc);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (cur.pix0[r] >= 0) {
                    const long long pix = cur.pix0[r] + (long long)(cls >> 1) * g.Win + (cls & 1);
                    dx[pix * 16 + i] = (cur.mk[cls][r] > 0.f) ? acc[r] : 0.f;
                }
            }
        }
        if (tn < ntiles) cur = nxt;
    }
}

inline bool geom_ok(const ConvGeom& g) {
    return g.C > 0 && g.Hin > 0 && g.Win > 0 && g.kh > 0 && g.kw > 0 && g.stride > 0 &&
           g.Ho == (g.Hin - g.kh) / g.stride + 1 && g.Wo == (g.Win - g.kw) / g.stride + 1 &&
           g.Ho > 0 && g.Wo > 0;
}

}  // namespace

extern "C" int smx_im2col_f32(const void* src, int32_t src_is_u8, int32_t channel_last, int64_t F,
                              int32_t C, int32_t Hin, int32_t Win, int32_t kh, int32_t kw,
                              int32_t stride, float scale_div, float* cols, smx_stream_t stream) {
    SMX_REQUIRE(src && cols, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, kh, kw, stride, (Hin - kh) / stride + 1, (Win - kw) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g), SMX_E_SHAPE);
    SMX_REQUIRE(!(src_is_u8 && channel_last), SMX_E_UNSUPPORTED);
    const long long total = (long long)F * g.Ho * g.Wo * C * kh * kw;
    const int blocks = grid_for(total);
    const bool small = total < (1ll << 31) && (long long)F * C * Hin * Win < (1ll << 31);
#define SMX_IM2COL(U8, CL)                                                                                    \
    do {                                                                                                      \
        if (small)                                                                                            \
            hipLaunchKernelGGL((im2col_kernel<U8, CL, unsigned>), dim3(blocks), dim3(256), 0, smx_s(stream),  \
                               src, g, total, scale_div, cols);                                               \
        else                                                                                                  \
            hipLaunchKernelGGL((im2col_kernel<U8, CL, long long>), dim3(blocks), dim3(256), 0, smx_s(stream), \
                               src, g, total, scale_div, cols);                                               \
    } while (0)
    if (src_is_u8) SMX_IM2COL(true, false);
    else if (channel_last) SMX_IM2COL(false, true);
    else SMX_IM2COL(false, false);
#undef SMX_IM2COL
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_conv_u8_forward_f32(const void* frames, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                                       int32_t stride, const float* W, const float* bias, int32_t cout, float* y,
                                       const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(frames && W && bias && y, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, k, k, stride, (Hin - k) / stride + 1, (Win - k) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g) && cout > 0, SMX_E_SHAPE);
    const int K = C * k * k;
    // dword gathers: four consecutive k's share a frame row and are 4-byte aligned
    if (cout > 16 || k % 4 || Win % 4 || stride % 4 || K % 16 || K > 256 || ((uintptr_t)frames & 3) ||
        ((uintptr_t)W & 15))
        return SMX_E_UNSUPPORTED;
    const long long rows = (long long)F * g.Ho * g.Wo;
    const long long ntiles = (rows + 15) >> 4;
    long long blocks = (ntiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;                   // 8 workgroups of 4 wavefronts per CU at most
    const unsigned char* fr = static_cast<const unsigned char*>(frames);
    void (*kern)(const unsigned char*, ConvGeom, long long, const float*, const float*, int, float*, const int*) =
        nullptr;
    switch (K / 16) {
        case 4: kern = conv_u8_fwd_kernel<4>; break;
        case 8: kern = conv_u8_fwd_kernel<8>; break;
        case 12: kern = conv_u8_fwd_kernel<12>; break;
        case 16: kern = conv_u8_fwd_kernel<16>; break;
        default: return SMX_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), fr, g, rows, W, bias, cout, y,
                       stop_flag);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int64_t smx_conv_u8_wgrad_ws_floats(int32_t cout, int32_t K) {
    return (int64_t)CONV_WGRAD_BLOCKS * ((int64_t)cout * K + cout);
}

extern "C" int smx_conv_u8_wgrad_f32(const void* frames, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                                     int32_t stride, const float* dy, int32_t cout, float* dW, float* db, float* ws,
                                     int64_t ws_floats, const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(frames && dy && dW && ws, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, k, k, stride, (Hin - k) / stride + 1, (Win - k) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g) && cout > 0, SMX_E_SHAPE);
    const int K = C * k * k;
    if (cout > 16 || k % 4 || Win % 4 || stride % 4 || K % 64 || K > 256 || ((uintptr_t)frames & 3))
        return SMX_E_UNSUPPORTED;
    SMX_REQUIRE(ws_floats >= smx_conv_u8_wgrad_ws_floats(cout, K), SMX_E_WORKSPACE);
    const long long rows = (long long)F * g.Ho * g.Wo;
    const long long ngroups = (rows + 15) >> 4;
    int blocks = (int)((ngroups + 31) / 32);             // >= 8 groups per wavefront: few partials for few rows
    if (blocks > CONV_WGRAD_BLOCKS) blocks = CONV_WGRAD_BLOCKS;
    if (blocks < 1) blocks = 1;
    float* part = ws;
    float* dbpart = ws + (size_t)CONV_WGRAD_BLOCKS * cout * K;
    const unsigned char* fr = static_cast<const unsigned char*>(frames);
    void (*kern)(const unsigned char*, ConvGeom, long long, const float*, int, float*, float*, const int*) = nullptr;
    switch (K / 16) {
        case 4: kern = conv_u8_wgrad_kernel<4>; break;
        case 8: kern = conv_u8_wgrad_kernel<8>; break;
        case 12: kern = conv_u8_wgrad_kernel<12>; break;
        case 16: kern = conv_u8_wgrad_kernel<16>; break;
        default: return SMX_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), fr, g, rows, dy, cout, part, dbpart,
                       stop_flag);
    SMX_LAUNCH_CHECK();
    const int n = cout * K + cout;
    hipLaunchKernelGGL(conv_partial_reduce_kernel, dim3((unsigned)((n + RED_EL - 1) / RED_EL)), dim3(256), 0, smx_s(stream),
                       part, dbpart, blocks, cout * K, cout, dW, db, stop_flag);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

static bool conv_cl_ok(const ConvGeom& g, int cout, const void* src) {
    const int NP = g.kh * g.kw;
    return g.C == 16 && cout <= 32 && (NP == 16 || NP == 9 || NP == 4) && (((uintptr_t)src) & 15) == 0;
}

extern "C" int smx_conv_cl_forward_f32(const float* src, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                                       int32_t stride, const float* W, const float* bias, int32_t cout, float* y,
                                       const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(src && W && bias && y, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, k, k, stride, (Hin - k) / stride + 1, (Win - k) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g) && cout > 0, SMX_E_SHAPE);
    if (!conv_cl_ok(g, cout, src)) return SMX_E_UNSUPPORTED;
    const long long rows = (long long)F * g.Ho * g.Wo;
    long long blocks = (((rows + 15) >> 4) + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const int NP = k * k, NT = (cout + 15) / 16;
    void (*kern)(const float*, ConvGeom, long long, const float*, const float*, int, float*, const int*) = nullptr;
    if (NP == 16) kern = NT == 1 ? conv_cl_fwd_kernel<16, 1> : conv_cl_fwd_kernel<16, 2>;
    else if (NP == 9) kern = NT == 1 ? conv_cl_fwd_kernel<9, 1> : conv_cl_fwd_kernel<9, 2>;
    else kern = NT == 1 ? conv_cl_fwd_kernel<4, 1> : conv_cl_fwd_kernel<4, 2>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), src, g, rows, W, bias, cout, y,
                       stop_flag);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int64_t smx_conv_cl_wgrad_ws_floats(int32_t cout, int32_t k) {
    const int64_t OP = 16 * ((cout + 15) / 16);
    return (int64_t)CONV_CL_WGRAD_BLOCKS * 4 * (OP * 16 * k * k + OP);
}

extern "C" int smx_conv_cl_wgrad_f32(const float* src, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                                     int32_t stride, const float* dy, int32_t cout, float* dW, float* db, float* ws,
                                     int64_t ws_floats, const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(src && dy && dW && ws, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, k, k, stride, (Hin - k) / stride + 1, (Win - k) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g) && cout > 0, SMX_E_SHAPE);
    if (!conv_cl_ok(g, cout, src)) return SMX_E_UNSUPPORTED;
    SMX_REQUIRE(ws_floats >= smx_conv_cl_wgrad_ws_floats(cout, k), SMX_E_WORKSPACE);
    const long long rows = (long long)F * g.Ho * g.Wo;
    const long long ngroups = (rows + 15) >> 4;
    int blocks = (int)((ngroups + 31) / 32);
    if (blocks > CONV_CL_WGRAD_BLOCKS) blocks = CONV_CL_WGRAD_BLOCKS;
    if (blocks < 1) blocks = 1;
    const int NP = k * k, NT = (cout + 15) / 16, OP = 16 * NT;
    float* part = ws;
    float* dbpart = ws + (size_t)CONV_CL_WGRAD_BLOCKS * 4 * OP * 16 * NP;
    void (*kern)(const float*, ConvGeom, long long, const float*, int, float*, float*, const int*) = nullptr;
    if (NP == 16) kern = NT == 1 ? conv_cl_wgrad_kernel<16, 1> : conv_cl_wgrad_kernel<16, 2>;
    else if (NP == 9) kern = NT == 1 ? conv_cl_wgrad_kernel<9, 1> : conv_cl_wgrad_kernel<9, 2>;
    else kern = NT == 1 ? conv_cl_wgrad_kernel<4, 1> : conv_cl_wgrad_kernel<4, 2>;
    const size_t lds = (size_t)4 * 16 * (16 * NP + 4) * sizeof(float);
    hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (er != hipSuccess) return (int)er;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, smx_s(stream), src, g, rows, dy, cout, part,
                       dbpart, stop_flag);
    SMX_LAUNCH_CHECK();
    const int n = OP * 16 * NP + OP;
    hipLaunchKernelGGL(conv_cl_wgrad_reduce_kernel, dim3((unsigned)((n + RED_EL - 1) / RED_EL)), dim3(256), 0, smx_s(stream),
                       part, dbpart, blocks * 4, OP, NP, cout, dW, db, stop_flag);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_conv_cl_dgrad_f32(const float* dy, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                                     int32_t stride, const float* W, int32_t cout, const float* relu_of, float* dx,
                                     const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(dy && W && dx, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, k, k, stride, (Hin - k) / stride + 1, (Win - k) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g) && cout > 0, SMX_E_SHAPE);
    if (C != 16 || cout > 32 || k != 2 * stride || (cout % 4 == 0 && (((uintptr_t)dy) & 15))) return SMX_E_UNSUPPORTED;
    const long long tiles = ((long long)F * Hin * Win / (stride * stride) + 15) >> 4;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    void (*kern)(const float*, ConvGeom, long long, const float*, int, const float*, float*, const int*) =
        cout <= 16 ? (cout % 4 == 0 ? conv_cl_dgrad_kernel<1, true> : conv_cl_dgrad_kernel<1, false>)
                   : (cout % 4 == 0 ? conv_cl_dgrad_kernel<2, true> : conv_cl_dgrad_kernel<2, false>);
    // stride 2, even maps (the reference's geometry): tiles of 2 x 2 cells, one gather for the four parity classes
    // (SMX_CONV_DGRAD_CLASSES=1 keeps the class-major kernel for A/B runs)
    static const bool classes = getenv("SMX_CONV_DGRAD_CLASSES") != nullptr;
    if (!classes && stride == 2 && Hin % 2 == 0 && Win % 2 == 0)
        kern = cout <= 16 ? (cout % 4 == 0 ? conv_cl_dgrad_cells_kernel<1, true> : conv_cl_dgrad_cells_kernel<1, false>)
                          : (cout % 4 == 0 ? conv_cl_dgrad_cells_kernel<2, true> : conv_cl_dgrad_cells_kernel<2, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), dy, g, (long long)F, W, cout, relu_of,
                       dx, stop_flag);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_col2im_f32(const float* dcols, int64_t F, int32_t C, int32_t Hin, int32_t Win,
                              int32_t kh, int32_t kw, int32_t stride, const float* relu_of,
                              float* dx, smx_stream_t stream) {
    SMX_REQUIRE(dcols && dx, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, kh, kw, stride, (Hin - kh) / stride + 1, (Win - kw) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g), SMX_E_SHAPE);
    const long long total = (long long)F * Hin * Win * C;
    if ((long long)F * g.Ho * g.Wo * C * kh * kw < (1ll << 31) && total < (1ll << 31))
        hipLaunchKernelGGL(col2im_kernel<unsigned>, dim3(grid_for(total)), dim3(256), 0, smx_s(stream), dcols, g,
                           total, relu_of, dx);
    else
        hipLaunchKernelGGL(col2im_kernel<long long>, dim3(grid_for(total)), dim3(256), 0, smx_s(stream), dcols, g,
                           total, relu_of, dx);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_flatten_order_f32(const float* in, int32_t O, int32_t C, int32_t P,
                                     int32_t to_channel_last, float* out, smx_stream_t stream) {
    SMX_REQUIRE(in && out, SMX_E_NULL);
    SMX_REQUIRE(O > 0 && C > 0 && P > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(flatten_order_kernel, dim3(grid_for((long long)O * C * P)), dim3(256), 0,
                       smx_s(stream), in, O, C, P, to_channel_last, out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
