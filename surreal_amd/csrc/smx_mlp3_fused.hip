// Fused z-filter + Linear-ReLU-Linear-ReLU-Linear[-Tanh] forward over every step of every
// sub-trajectory: the critic pass of PPOLearner._gae_and_return (surreal/learner/ppo.py:376-386
// -> surreal/model/ppo_net.py:284-315 -> model_builders/builders.py:159-175), and the actor
// pass where it runs over all steps.
//
// CDNA4 design (not a tiled-GEMM translation):
//   * everything is computed TRANSPOSED: h^T = W . x^T, so the MFMA "N" axis (lane & 31) is the
//     data row and the MFMA "M" axis is the output feature.  The C/D fragment of
//     v_mfma_f32_32x32x2_f32 then holds, per lane, one data row and 16 features -- exactly the
//     B-operand shape of the NEXT layer's MFMA.  Activations therefore never leave registers:
//     layer-1 accumulators (after bias+ReLU) are fed straight back as B operands of layer 2, and
//     layer 2's into layer 3.  One wavefront owns 32 data rows end to end.
//   * the K index inside a 32-wide chunk is permuted (step 4q+r, half kh -> k = 8q+4kh+r) so
//     that each lane's four consecutive MFMA steps read ONE 16-byte LDS word (ds_read_b128),
//     and so that the C-fragment register order IS the k order of the next layer.
//   * weights stream HBM/L2 -> registers -> LDS in K-chunks of 32 (double buffered, one barrier
//     per chunk); x is read once from HBM with full 128-byte lines, z-filtered in registers on
//     the way into LDS.  Rows of 36 floats (144 B) make every ds_read_b128 conflict-free.
//   * FP32 MFMA (exact fp32, no bf16/xf32): 1e-5 parity with the CPU reference is the contract.
#include "smx_common.h"
#include "smx_mlp3_fused.inc.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int ROWS_PER_WG = 128;  // 4 waves x 32 data rows
constexpr int LDS_STRIDE = 36;    // floats per staged row (32 + 4 pad)

inline bool pick_variant(int H1, int H2, int* nt1, int* nt2) {
    if (H1 <= 64 && H2 <= 64) { *nt1 = 2; *nt2 = 2; return true; }
    if (H1 <= 320 && H2 <= 224) { *nt1 = 10; *nt2 = 7; return true; }
    return false;
}

__device__ __forceinline__ float zf(float x, float m, float s) {
    float v = (x - m) / s;                              // z_filter.py:77
    if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);
    return v;
}

__device__ __forceinline__ float relu_f(float v) { return (v < 0.f) ? 0.f : v; }

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)


// stage NT*32 rows x 32 floats: thread -> rows (srow + 32 i), 16 B at sk8
template <int NT>
__device__ __forceinline__ void load_w(float4 (&wreg)[NT], const float* chunk, int srow, int sk8) {
    const float4* src = reinterpret_cast<const float4*>(chunk);
#pragma unroll
    for (int i = 0; i < NT; ++i) wreg[i] = src[(srow + 32 * i) * 8 + sk8];
}
template <int NT>
__device__ __forceinline__ void store_w(const float4 (&wreg)[NT], float* Wb, int srow, int sk8) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
        *reinterpret_cast<float4*>(Wb + (srow + 32 * i) * LDS_STRIDE + 4 * sk8) = wreg[i];
}

// Vector x staging (D % 4 == 0, 16-byte aligned rows): the loads are UNCONDITIONAL -- a k past D
// or a row past the batch reads a clamped, valid address and is zeroed when it is written to LDS.
// A load guarded by a branch (or followed by a select) makes hipcc wait for it on the spot, which
// put a full HBM round trip at the top of every K chunk.
template <bool HASZ>
__device__ __forceinline__ void load_x_vec(float4 (&xreg)[4], float4& zm, float4& zs,
                                           const float* const (&xp)[4], const float* zmean,
                                           const float* zstd, int k0, int D) {
    const int kc = (k0 < D) ? k0 : D - 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) xreg[i] = *reinterpret_cast<const float4*>(xp[i] + kc);
    if (HASZ) {
        zm = *reinterpret_cast<const float4*>(zmean + kc);
        zs = *reinterpret_cast<const float4*>(zstd + kc);
    }
}

template <bool HASZ>
__device__ __forceinline__ void store_x_vec(float* Xb, const float4 (&xreg)[4], const float4 zm,
                                            const float4 zs, const bool (&xok)[4], int k0, int D,
                                            int srow, int sk4) {
    const bool kin = k0 < D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 v = xreg[i];
        const bool ok = xok[i] && kin;          // padded k (>= D) and padded rows stay exactly zero
        if (HASZ) {      // same arithmetic as stage_x below: (x - m) * (1 / s)
            float a = (v.x - zm.x) * (1.0f / zs.x), b = (v.y - zm.y) * (1.0f / zs.y);
            float c = (v.z - zm.z) * (1.0f / zs.z), d = (v.w - zm.w) * (1.0f / zs.w);
            if (a == a) a = fminf(fmaxf(a, -5.0f), 5.0f);
            if (b == b) b = fminf(fmaxf(b, -5.0f), 5.0f);
            if (c == c) c = fminf(fmaxf(c, -5.0f), 5.0f);
            if (d == d) d = fminf(fmaxf(d, -5.0f), 5.0f);
            v.x = ok ? a : 0.f;
            v.y = ok ? b : 0.f;
            v.z = ok ? c : 0.f;
            v.w = ok ? d : 0.f;
        } else if (!ok) {
            v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *reinterpret_cast<float4*>(Xb + (srow + 32 * i) * LDS_STRIDE + sk4) = v;
    }
}

// (x - m) * (1/s) instead of (x - m) / s: one rounding more than z_filter.py:77 (<= 1 ulp of the
// filtered input, ~1e-7 relative -- far inside the 1e-5 contract) for a quarter of the vector-ALU
// work: the reciprocal is formed once per lane and chunk, not once per element.
__device__ __forceinline__ float zf_mul(float x, float m, float r) {
    float v = (x - m) * r;
    if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);
    return v;
}

template <bool HASZ>
__device__ __forceinline__ void stage_x(float* dst, float4 v, const float4 zm, const float4 rz,
                                        const bool ok) {
    if (HASZ) {
        // computed unconditionally, then selected: a conditional term becomes a divergent branch
        const float a = zf_mul(v.x, zm.x, rz.x), b = zf_mul(v.y, zm.y, rz.y);
        const float c = zf_mul(v.z, zm.z, rz.z), d = zf_mul(v.w, zm.w, rz.w);
        v.x = ok ? a : 0.f;
        v.y = ok ? b : 0.f;
        v.z = ok ? c : 0.f;
        v.w = ok ? d : 0.f;
    } else if (!ok) {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    *reinterpret_cast<float4*>(dst) = v;
}

// XMODE 0: generic x staging (any D / alignment, run-time z-filter switch); 1: vector staging with
// the z-filter; 2: vector staging without it.
template <int NT1, int NT2, bool OUT1, int XMODE>
__global__ __launch_bounds__(256, 1) void mlp3_fused_kernel(FusedArgs A) {
    constexpr int WR = (NT1 > NT2 ? NT1 : NT2) * 32;  // rows of one weight staging buffer
    extern __shared__ float lds[];
    float* Wb0 = lds;
    float* Wb1 = Wb0 + WR * LDS_STRIDE;
    float* Xb0 = Wb1 + WR * LDS_STRIDE;
    float* Xb1 = Xb0 + ROWS_PER_WG * LDS_STRIDE;
    float* b1s = Xb1 + ROWS_PER_WG * LDS_STRIDE;
    float* b2s = b1s + NT1 * 32;
    float* w3s = b2s + NT2 * 32;  // OUT1: row 0 of W3 (NT2*32) ; generic: b3 (32)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const long row0 = (long)blockIdx.x * ROWS_PER_WG;
    const PackLayout L = pack_layout(NT1, NT2, A.KC1);
    const float* W1p = A.packed + L.w1;
    const float* W2p = A.packed + L.w2;
    const float* W3p = A.packed + L.w3;

    // ---- staging geometry: thread -> (row srow + 32 i, 4 consecutive k at sk4) ----------
    const int srow = tid >> 3, sk8 = tid & 7, sk4 = sk8 * 4;
    const float* xp[4];
    bool xok[4];
    {
        const int T = A.T0 + A.T1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = row0 + srow + 32 * i;
            xok[i] = r < A.total_rows;
            const long rr = xok[i] ? r : 0;
            const long g = rr / T;
            const int tt = (int)(rr - g * T);
            xp[i] = (tt < A.T0) ? A.x_main + (g * A.T0 + tt) * (long)A.D
                                : A.x_tail + (g * A.T1 + (tt - A.T0)) * (long)A.D;
        }
    }
    for (int i = tid; i < NT1 * 32; i += 256) b1s[i] = A.packed[L.b1 + i];
    for (int i = tid; i < NT2 * 32; i += 256) b2s[i] = A.packed[L.b2 + i];
    if (OUT1) {
        for (int i = tid; i < NT2 * 32; i += 256) w3s[i] = W3p[(size_t)(i >> 5) * 1024 + (i & 31)];
    } else {
        if (tid < 32) w3s[tid] = A.packed[L.b3 + tid];
    }

    // x staging is split in two so that the HBM latency hides under the MFMA section:
    // load_x issues the global loads (raw values) before the compute, store_x applies the
    // z-filter and writes LDS after it.
    float4 xreg[4], zmreg, zsreg;
    const bool blk_full = (row0 + ROWS_PER_WG <= A.total_rows);
    auto load_x = [&](int c) {
        const int k0 = 32 * c + sk4;
        if (blk_full && A.xvec && 32 * c + 32 <= A.D) {  // block-uniform fast path
#pragma unroll
            for (int i = 0; i < 4; ++i) xreg[i] = *reinterpret_cast<const float4*>(xp[i] + k0);
            if (A.zmean) {
                zmreg = *reinterpret_cast<const float4*>(A.zmean + k0);
                zsreg = *reinterpret_cast<const float4*>(A.zstd + k0);
            }
        } else {
            float zm[4] = {0.f, 0.f, 0.f, 0.f}, zs[4] = {1.f, 1.f, 1.f, 1.f};
            if (A.zmean) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k0 + e < A.D) { zm[e] = A.zmean[k0 + e]; zs[e] = A.zstd[k0 + e]; }
            }
            zmreg = make_float4(zm[0], zm[1], zm[2], zm[3]);
            zsreg = make_float4(zs[0], zs[1], zs[2], zs[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (xok[i]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + e < A.D) v[e] = xp[i][k0 + e];
                }
                xreg[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    auto store_x = [&](float* Xb, int c) {
        const int k0 = 32 * c + sk4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = xreg[i];
            if (A.zmean) {
                // padded k (>= D) and padded rows must stay exactly zero
                v.x = (xok[i] && k0 + 0 < A.D) ? zf(v.x, zmreg.x, zsreg.x) : 0.f;
                v.y = (xok[i] && k0 + 1 < A.D) ? zf(v.y, zmreg.y, zsreg.y) : 0.f;
                v.z = (xok[i] && k0 + 2 < A.D) ? zf(v.z, zmreg.z, zsreg.z) : 0.f;
                v.w = (xok[i] && k0 + 3 < A.D) ? zf(v.w, zmreg.w, zsreg.w) : 0.f;
            }
            *reinterpret_cast<float4*>(Xb + (srow + 32 * i) * LDS_STRIDE + sk4) = v;
        }
    };

    // ======================= layer 1: acc1[t] = W1[tile t] . x^T ===========================
    f32x16 acc1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s) acc1[t][s] = 0.f;

    float4 w1reg[NT1];
    load_w<NT1>(w1reg, W1p, srow, sk8);
    if (XMODE == 0) load_x(0);
    else load_x_vec<XMODE == 1>(xreg, zmreg, zsreg, xp, A.zmean, A.zstd, sk4, A.D);
    store_w<NT1>(w1reg, Wb0, srow, sk8);
    if (XMODE == 0) store_x(Xb0, 0);
    else store_x_vec<XMODE == 1>(Xb0, xreg, zmreg, zsreg, xok, sk4, A.D, srow, sk4);
    __syncthreads();

    for (int c = 0; c < A.KC1; ++c) {
        const float* Wc = (c & 1) ? Wb1 : Wb0;
        const float* Xc = (c & 1) ? Xb1 : Xb0;
        float* Wn = (c & 1) ? Wb0 : Wb1;
        float* Xn = (c & 1) ? Xb0 : Xb1;
        // branch-free prefetch: the last iteration re-stages its own chunk (never read again)
        const int cn = (c + 1 < A.KC1) ? c + 1 : c;
        const float* xrow = Xc + (wv * 32 + j) * LDS_STRIDE + 4 * kh;
        const float* wrow = Wc + j * LDS_STRIDE + 4 * kh;
        // one 8-wide k group of the chunk: 11 LDS fragment reads, 4 x NT1 MFMAs
        auto kgroup = [&](int q) {
            const float4 b = *reinterpret_cast<const float4*>(xrow + 8 * q);
            float4 a[NT1];
#pragma unroll
            for (int t = 0; t < NT1; ++t)
                a[t] = *reinterpret_cast<const float4*>(wrow + t * 32 * LDS_STRIDE + 8 * q);
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc1[t] = MFMA32(a[t].x, b.x, acc1[t]);
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc1[t] = MFMA32(a[t].y, b.y, acc1[t]);
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc1[t] = MFMA32(a[t].z, b.z, acc1[t]);
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc1[t] = MFMA32(a[t].w, b.w, acc1[t]);
        };
        if (XMODE == 0) {
            load_w<NT1>(w1reg, W1p + (size_t)cn * NT1 * 1024, srow, sk8);
            load_x(cn);
            kgroup(0); kgroup(1); kgroup(2); kgroup(3);
            store_w<NT1>(w1reg, Wn, srow, sk8);
            store_x(Xn, cn);
        } else {
            // Software pipeline of the next chunk's staging in two batches of 8 x 16 B per lane:
            //   issue A (x rows, z-filter stats, weight tiles 0-1) | k groups 0, 1 | A -> LDS,
            //   issue B (weight tiles 2..NT1-1)                    | k groups 2, 3 | B -> LDS.
            // Each batch has two k groups (~2.4 us of MFMA issue) to land, and only one batch of
            // staging registers is live at a time.  The sched_barriers pin that order: left
            // alone, hipcc sinks the loads behind the MFMA block and waits for each in front of
            // its LDS store.
            // (named scalars, not arrays: an array that lives across a sched_barrier is left in
            // scratch memory by hipcc)
            static_assert(XMODE == 0 || NT1 == 10, "vector staging is written for 10 feature tiles");
            const float4* wsrc = reinterpret_cast<const float4*>(W1p + (size_t)cn * NT1 * 1024);
            const int k0n = 32 * cn + sk4;
            const int kc = (k0n < A.D) ? k0n : A.D - 4;
            const bool kin = k0n < A.D;
#define SMX_WLD(i) wsrc[(srow + 32 * (i)) * 8 + sk8]
#define SMX_WST(i, v) *reinterpret_cast<float4*>(Wn + (srow + 32 * (i)) * LDS_STRIDE + 4 * sk8) = (v)
            float4 x0 = *reinterpret_cast<const float4*>(xp[0] + kc);
            float4 x1 = *reinterpret_cast<const float4*>(xp[1] + kc);
            float4 x2 = *reinterpret_cast<const float4*>(xp[2] + kc);
            float4 x3 = *reinterpret_cast<const float4*>(xp[3] + kc);
            float4 zm = make_float4(0.f, 0.f, 0.f, 0.f), zs = make_float4(1.f, 1.f, 1.f, 1.f);
            if (XMODE == 1) {
                zm = *reinterpret_cast<const float4*>(A.zmean + kc);
                zs = *reinterpret_cast<const float4*>(A.zstd + kc);
            }
            const float4 wa0 = SMX_WLD(0), wa1 = SMX_WLD(1);
            __builtin_amdgcn_sched_barrier(0);
            kgroup(0); kgroup(1);
            __builtin_amdgcn_sched_barrier(0);
            const float4 wb2 = SMX_WLD(2), wb3 = SMX_WLD(3), wb4 = SMX_WLD(4), wb5 = SMX_WLD(5);
            const float4 wb6 = SMX_WLD(6), wb7 = SMX_WLD(7), wb8 = SMX_WLD(8), wb9 = SMX_WLD(9);
            __builtin_amdgcn_sched_barrier(0);
            // batch A: the z-filter divisions may run under these MFMAs; its LDS stores follow the
            // fragment reads in program order (hipcc cannot tell the two staging buffers apart).
            // The empty asm re-defines the loaded x registers HERE: without it the (pure) z-filter
            // arithmetic is emitted right behind the loads at the top of the chunk, in front of the
            // first sched_barrier, and waits for HBM there.
            asm volatile("" : "+v"(x0.x), "+v"(x0.y), "+v"(x0.z), "+v"(x0.w), "+v"(x1.x), "+v"(x1.y),
                              "+v"(x1.z), "+v"(x1.w), "+v"(x2.x), "+v"(x2.y), "+v"(x2.z), "+v"(x2.w),
                              "+v"(x3.x), "+v"(x3.y), "+v"(x3.z), "+v"(x3.w));
            asm volatile("" : "+v"(zs.x), "+v"(zs.y), "+v"(zs.z), "+v"(zs.w));
            const float4 rz = make_float4(1.0f / zs.x, 1.0f / zs.y, 1.0f / zs.z, 1.0f / zs.w);
            kgroup(2); kgroup(3);
            stage_x<XMODE == 1>(Xn + (srow + 0) * LDS_STRIDE + sk4, x0, zm, rz, xok[0] && kin);
            stage_x<XMODE == 1>(Xn + (srow + 32) * LDS_STRIDE + sk4, x1, zm, rz, xok[1] && kin);
            stage_x<XMODE == 1>(Xn + (srow + 64) * LDS_STRIDE + sk4, x2, zm, rz, xok[2] && kin);
            stage_x<XMODE == 1>(Xn + (srow + 96) * LDS_STRIDE + sk4, x3, zm, rz, xok[3] && kin);
            SMX_WST(0, wa0); SMX_WST(1, wa1);
            __builtin_amdgcn_sched_barrier(0);
            SMX_WST(2, wb2); SMX_WST(3, wb3); SMX_WST(4, wb4); SMX_WST(5, wb5);
            SMX_WST(6, wb6); SMX_WST(7, wb7); SMX_WST(8, wb8); SMX_WST(9, wb9);
#undef SMX_WLD
#undef SMX_WST
        }
        __syncthreads();
    }

    // bias + ReLU in the C-fragment layout: reg s of tile t holds feature 32t + 8(s>>2) + 4kh + (s&3)
#pragma unroll
    for (int t = 0; t < NT1; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = *reinterpret_cast<const float4*>(b1s + 32 * t + 8 * g + 4 * kh);
            acc1[t][4 * g + 0] = relu_f(acc1[t][4 * g + 0] + bb.x);
            acc1[t][4 * g + 1] = relu_f(acc1[t][4 * g + 1] + bb.y);
            acc1[t][4 * g + 2] = relu_f(acc1[t][4 * g + 2] + bb.z);
            acc1[t][4 * g + 3] = relu_f(acc1[t][4 * g + 3] + bb.w);
        }

    // ======================= layer 2: acc2[u] = W2[tile u] . h1^T ==========================
    f32x16 acc2[NT2];
#pragma unroll
    for (int u = 0; u < NT2; ++u)
#pragma unroll
        for (int s = 0; s < 16; ++s) acc2[u][s] = 0.f;

    float4 w2reg[NT2];
    load_w<NT2>(w2reg, W2p, srow, sk8);
    store_w<NT2>(w2reg, Wb0, srow, sk8);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const float* Wc = (t & 1) ? Wb1 : Wb0;
        float* Wn = (t & 1) ? Wb0 : Wb1;
        const float* wnext = W2p + (size_t)((t + 1 < NT1) ? t + 1 : t) * NT2 * 1024;
        const float* wrow = Wc + j * LDS_STRIDE + 4 * kh;
        auto chunk2 = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 a[NT2];
#pragma unroll
                for (int u = 0; u < NT2; ++u)
                    a[u] = *reinterpret_cast<const float4*>(wrow + u * 32 * LDS_STRIDE + 8 * q);
#pragma unroll
                for (int u = 0; u < NT2; ++u) acc2[u] = MFMA32(a[u].x, acc1[t][4 * q + 0], acc2[u]);
#pragma unroll
                for (int u = 0; u < NT2; ++u) acc2[u] = MFMA32(a[u].y, acc1[t][4 * q + 1], acc2[u]);
#pragma unroll
                for (int u = 0; u < NT2; ++u) acc2[u] = MFMA32(a[u].z, acc1[t][4 * q + 2], acc2[u]);
#pragma unroll
                for (int u = 0; u < NT2; ++u) acc2[u] = MFMA32(a[u].w, acc1[t][4 * q + 3], acc2[u]);
            }
        };
        if (XMODE == 0) {
            load_w<NT2>(w2reg, wnext, srow, sk8);
            chunk2();
            store_w<NT2>(w2reg, Wn, srow, sk8);
        } else {
            // same pinning as layer 1: the next chunk's weight loads are issued in front of the
            // MFMAs (named scalars: see above)
            static_assert(XMODE == 0 || NT2 == 7, "vector staging is written for 7 feature tiles");
            const float4* wsrc = reinterpret_cast<const float4*>(wnext);
#define SMX_WLD(i) wsrc[(srow + 32 * (i)) * 8 + sk8]
#define SMX_WST(i, v) *reinterpret_cast<float4*>(Wn + (srow + 32 * (i)) * LDS_STRIDE + 4 * sk8) = (v)
            const float4 v0 = SMX_WLD(0), v1 = SMX_WLD(1), v2 = SMX_WLD(2), v3 = SMX_WLD(3);
            const float4 v4 = SMX_WLD(4), v5 = SMX_WLD(5), v6 = SMX_WLD(6);
            __builtin_amdgcn_sched_barrier(0);
            chunk2();
            __builtin_amdgcn_sched_barrier(0);
            SMX_WST(0, v0); SMX_WST(1, v1); SMX_WST(2, v2); SMX_WST(3, v3);
            SMX_WST(4, v4); SMX_WST(5, v5); SMX_WST(6, v6);
#undef SMX_WLD
#undef SMX_WST
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NT2; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = *reinterpret_cast<const float4*>(b2s + 32 * u + 8 * g + 4 * kh);
            acc2[u][4 * g + 0] = relu_f(acc2[u][4 * g + 0] + bb.x);
            acc2[u][4 * g + 1] = relu_f(acc2[u][4 * g + 1] + bb.y);
            acc2[u][4 * g + 2] = relu_f(acc2[u][4 * g + 2] + bb.z);
            acc2[u][4 * g + 3] = relu_f(acc2[u][4 * g + 3] + bb.w);
        }

    // ======================= layer 3 ========================================================
    const long myrow = row0 + wv * 32 + j;
    if (OUT1) {
        // one output: a 2*NT2*16-term dot product per lane pair, on the VALU
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < NT2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 ww = *reinterpret_cast<const float4*>(w3s + 32 * u + 8 * g + 4 * kh);
                v = fmaf(acc2[u][4 * g + 0], ww.x, v);
                v = fmaf(acc2[u][4 * g + 1], ww.y, v);
                v = fmaf(acc2[u][4 * g + 2], ww.z, v);
                v = fmaf(acc2[u][4 * g + 3], ww.w, v);
            }
        v += __shfl_xor(v, 32, 64);
        v += A.packed[L.b3];
        if (A.out_act == SMX_ACT_TANH) v = tanhf(v);
        if (kh == 0 && myrow < A.total_rows) A.out[myrow] = v;
    } else {
        // OUT <= 32 outputs: one more MFMA tile; all of W3 (NT2 chunks) fits one staging buffer
        f32x16 acc3;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc3[s] = 0.f;
        {
            const float4* src = reinterpret_cast<const float4*>(W3p);
#pragma unroll
            for (int u = 0; u < NT2; ++u) {
                const float4 t4 = src[(u * 32 + srow) * 8 + sk8];
                *reinterpret_cast<float4*>(Wb0 + (u * 32 + srow) * LDS_STRIDE + sk4) = t4;
            }
        }
        __syncthreads();
        const float* wrow = Wb0 + j * LDS_STRIDE + 4 * kh;
#pragma unroll
        for (int u = 0; u < NT2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(wrow + u * 32 * LDS_STRIDE + 8 * q);
                acc3 = MFMA32(a.x, acc2[u][4 * q + 0], acc3);
                acc3 = MFMA32(a.y, acc2[u][4 * q + 1], acc3);
                acc3 = MFMA32(a.z, acc2[u][4 * q + 2], acc3);
                acc3 = MFMA32(a.w, acc2[u][4 * q + 3], acc3);
            }
        if (myrow < A.total_rows) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int o = 8 * (s >> 2) + 4 * kh + (s & 3);
                if (o < A.OUT) {
                    float v = acc3[s] + w3s[o];
                    if (A.out_act == SMX_ACT_TANH) v = tanhf(v);
                    else if (A.out_act == SMX_ACT_RELU) v = relu_f(v);
                    A.out[myrow * A.OUT + o] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// weight repacking: [KC][NT*32][32] K-chunked, zero padded
// ---------------------------------------------------------------------------
// optional rider: the z-filter statistics of the pass that follows (smx_zfilter_stats_f32's arithmetic,
// z_filter.py:74-76) -- the repack and the statistics are the two launches in front of every critic pass
struct ZStatsRider {
    const float *rs, *rsq, *cnt;
    float *mean, *stdv;
    int D;
    float eps;
};

__global__ __launch_bounds__(256) void mlp3_pack_kernel(smx_mlp3_t net, float* __restrict__ packed,
                                                        int NT1, int NT2, int KC1, ZStatsRider Z) {
    if (Z.rs) {
        for (int k = blockIdx.x * 256 + threadIdx.x; k < Z.D; k += gridDim.x * 256) {
            const float c = Z.cnt[0];
            const float m = Z.rs[k] / c;                // z_filter.py:74
            const float var = Z.rsq[k] / c - m * m;     // z_filter.py:75
            float sd = sqrtf(var);                      // .pow(0.5): NaN for var < 0, like torch
            if (sd == sd) sd = fmaxf(sd, Z.eps);        // torch.clamp(min=eps) propagates NaN
            Z.mean[k] = m;
            Z.stdv[k] = sd;
        }
    }
    const PackLayout L = pack_layout(NT1, NT2, KC1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < L.total; i += (size_t)gridDim.x * 256) {
        float v = 0.f;
        if (i < L.b1) {
            const size_t e = i - L.w1;
            const int kk = (int)(e & 31);
            const size_t rc = e >> 5;
            const int f = (int)(rc % (size_t)(NT1 * 32)), c = (int)(rc / (size_t)(NT1 * 32));
            const int k = 32 * c + kk;
            if (f < net.H1 && k < net.D) v = net.W1[(size_t)f * net.D + k];
        } else if (i < L.w2) {
            const int f = (int)(i - L.b1);
            if (f < net.H1) v = net.b1[f];
        } else if (i < L.b2) {
            const size_t e = i - L.w2;
            const int kk = (int)(e & 31);
            const size_t rc = e >> 5;
            const int f = (int)(rc % (size_t)(NT2 * 32)), c = (int)(rc / (size_t)(NT2 * 32));
            const int k = 32 * c + kk;
            if (f < net.H2 && k < net.H1) v = net.W2[(size_t)f * net.H1 + k];
        } else if (i < L.w3) {
            const int f = (int)(i - L.b2);
            if (f < net.H2) v = net.b2[f];
        } else if (i < L.b3) {
            const size_t e = i - L.w3;
            const int kk = (int)(e & 31);
            const size_t rc = e >> 5;
            const int o = (int)(rc & 31), c = (int)(rc >> 5);
            const int k = 32 * c + kk;
            if (o < net.OUT && k < net.H2) v = net.W3[(size_t)o * net.H2 + k];
        } else {
            const int o = (int)(i - L.b3);
            if (o < net.OUT) v = net.b3[o];
        }
        packed[i] = v;
    }
}

template <int NT1, int NT2>
int launch_fused(const FusedArgs& A, hipStream_t st) {
    constexpr int WR = (NT1 > NT2 ? NT1 : NT2) * 32;
    const size_t lds = (size_t)(2 * WR * LDS_STRIDE + 2 * ROWS_PER_WG * LDS_STRIDE + NT1 * 32 +
                                NT2 * 32 + NT2 * 32) * sizeof(float);
    const unsigned grid = (unsigned)((A.total_rows + ROWS_PER_WG - 1) / ROWS_PER_WG);
    // vector x staging only in the large-tile build (the small one is not a throughput path)
    const int xmode = (NT1 > 2 && A.xvec && A.D >= 4) ? (A.zmean ? 1 : 2) : 0;
    void (*k)(FusedArgs) = nullptr;
    if (A.OUT == 1) {
        k = xmode == 1 ? mlp3_fused_kernel<NT1, NT2, true, (NT1 > 2 ? 1 : 0)>
          : xmode == 2 ? mlp3_fused_kernel<NT1, NT2, true, (NT1 > 2 ? 2 : 0)>
                       : mlp3_fused_kernel<NT1, NT2, true, 0>;
    } else {
        k = xmode == 1 ? mlp3_fused_kernel<NT1, NT2, false, (NT1 > 2 ? 1 : 0)>
          : xmode == 2 ? mlp3_fused_kernel<NT1, NT2, false, (NT1 > 2 ? 2 : 0)>
                       : mlp3_fused_kernel<NT1, NT2, false, 0>;
    }
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, A);
    e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}

}  // namespace

static int g_exact_zfilter = 0;
// 1: z-filter with the reference's division (z_filter.py:77) instead of (x - m) * (1 / s): the generic staging
// path of the 32-row kernel.  Process-wide; the goldens run with 0.
extern "C" int smx_mlp3_fused_exact_zfilter(int32_t on) {
    g_exact_zfilter = on ? 1 : 0;
    return SMX_OK;
}

static long long* g_fused_tbuf = nullptr;
// timing builds (-DSMX_FUSED_TIMING, scripts/bench_fused.py): where the 16-row kernel writes its phase timestamps
extern "C" void smx_mlp3_fused_debug_tbuf(void* p) { g_fused_tbuf = (long long*)p; }

extern "C" size_t smx_mlp3_packed_bytes(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    int nt1, nt2;
    if (D <= 0 || OUT <= 0 || OUT > 32 || !pick_variant(H1, H2, &nt1, &nt2)) return 0;
    return pack_layout(nt1, nt2, (D + 31) / 32).total * sizeof(float);
}

static int pack_launch(const smx_mlp3_t* net, float* packed, size_t packed_bytes, const ZStatsRider& Z,
                       smx_stream_t stream) {
    SMX_REQUIRE(net && packed && net->W1 && net->b1 && net->W2 && net->b2 && net->W3 && net->b3,
                SMX_E_NULL);
    SMX_REQUIRE(net->D > 0 && net->H1 > 0 && net->H2 > 0 && net->OUT > 0, SMX_E_SHAPE);
    int nt1, nt2;
    SMX_REQUIRE(net->OUT <= 32 && pick_variant(net->H1, net->H2, &nt1, &nt2), SMX_E_UNSUPPORTED);
    const int KC1 = (net->D + 31) / 32;
    const PackLayout L = pack_layout(nt1, nt2, KC1);
    SMX_REQUIRE(packed_bytes >= L.total * sizeof(float), SMX_E_WORKSPACE);
    SMX_REQUIRE(((uintptr_t)packed & 15) == 0, SMX_E_ALIGN);
    unsigned blocks = (unsigned)((L.total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(mlp3_pack_kernel, dim3(blocks), dim3(256), 0, smx_s(stream), *net, packed, nt1,
                       nt2, KC1, Z);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_mlp3_pack_f32(const smx_mlp3_t* net, float* packed, size_t packed_bytes,
                                 smx_stream_t stream) {
    ZStatsRider Z;
    memset(&Z, 0, sizeof(Z));
    return pack_launch(net, packed, packed_bytes, Z, stream);
}

extern "C" int smx_mlp3_pack_zstats_f32(const smx_mlp3_t* net, float* packed, size_t packed_bytes,
                                        const float* running_sum, const float* running_sumsq,
                                        const float* count, int32_t D, float eps, float* mean_out,
                                        float* std_out, smx_stream_t stream) {
    SMX_REQUIRE(running_sum && running_sumsq && count && mean_out && std_out, SMX_E_NULL);
    SMX_REQUIRE(D > 0, SMX_E_SHAPE);
    ZStatsRider Z;
    Z.rs = running_sum; Z.rsq = running_sumsq; Z.cnt = count; Z.mean = mean_out; Z.stdv = std_out;
    Z.D = D; Z.eps = eps;
    return pack_launch(net, packed, packed_bytes, Z, stream);
}

extern "C" int smx_mlp3_forward_fused_f32(const float* packed, int32_t D, int32_t H1, int32_t H2,
                                          int32_t OUT, const float* x_main, const float* x_tail,
                                          int64_t G, int32_t T0, int32_t T1, const float* zmean,
                                          const float* zstd, float* out, int32_t out_act,
                                          smx_stream_t stream) {
    SMX_REQUIRE(packed && x_main && out, SMX_E_NULL);
    SMX_REQUIRE(T1 == 0 || x_tail, SMX_E_NULL);
    SMX_REQUIRE((zmean == nullptr) == (zstd == nullptr), SMX_E_NULL);
    SMX_REQUIRE(G > 0 && T0 > 0 && T1 >= 0 && D > 0 && OUT > 0, SMX_E_SHAPE);
    int nt1, nt2;
    SMX_REQUIRE(OUT <= 32 && pick_variant(H1, H2, &nt1, &nt2), SMX_E_UNSUPPORTED);
    SMX_REQUIRE(((uintptr_t)packed & 15) == 0, SMX_E_ALIGN);
    FusedArgs A;
    A.packed = packed;
    A.x_main = x_main;
    A.x_tail = x_tail ? x_tail : x_main;
    A.zmean = zmean;
    A.zstd = zstd;
    A.out = out;
    A.total_rows = (long)G * (T0 + T1);
    A.T0 = T0;
    A.T1 = T1;
    A.D = D;
    A.OUT = OUT;
    A.KC1 = (D + 31) / 32;
    A.out_act = out_act;
    A.tbuf = g_fused_tbuf;
    A.exp = 0;
    A.h1_out = A.h2_out = nullptr;
    A.H1 = H1; A.H2 = H2; A.out_ld = OUT;
    A.stop = nullptr;
    A.xvec = (D % 4 == 0) && (((uintptr_t)x_main & 15) == 0) &&
             (T1 == 0 || ((uintptr_t)x_tail & 15) == 0) &&
             (zmean == nullptr || ((((uintptr_t)zmean | (uintptr_t)zstd) & 15) == 0));
    if (nt1 == 2) return launch_fused<2, 2>(A, smx_s(stream));
    // 16-row wavefronts (smx_mlp3_rows16.hip) where the shape allows; SMX_FUSED32=1 keeps the 32-row
    // kernel for A/B measurements (scripts/bench_gemm.py)
    static const bool force32 = getenv("SMX_FUSED32") != nullptr;
    if (g_exact_zfilter && zmean) A.xvec = 0;
    if (!force32 && A.xvec) {
        const int rc = smx_rows16_launch(A, H1, H2, smx_s(stream));
        if (rc != SMX_E_UNSUPPORTED) return rc;
    }
    return launch_fused<10, 7>(A, smx_s(stream));
}

// ---------------------------------------------------------------------------------------------
// smx_mlp3_forward_rows_f32: smx_mlp3_forward_f32 (the forward pass that KEEPS h1 / h2 for the backward) over many
// rows -- the MLPs on top of an LSTM / CNN stem, B*E ~ 10^5 rows per epoch (surreal/model/ppo_net.py:284-315 under
// surreal/learner/ppo.py:227-353).  One launch of the 16-row fused kernel instead of three layer GEMMs: x is read
// once, the activations go from the accumulators to memory once (and on, in registers, to the next layer).  The
// weights are repacked into `packed` first (they change every epoch): one short launch.
// SMX_E_UNSUPPORTED (the caller then uses the layered smx_mlp3_forward_f32) outside the fused kernel's fast path:
// D % 4, H1 % 4, H2 % 4 == 0, 64 < H1 <= 320, 64 < H2 <= 224, OUT <= 32, 16-byte aligned x / h1 / h2.
// ---------------------------------------------------------------------------------------------
extern "C" int32_t smx_mlp3_forward_rows_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    int nt1, nt2;
    return D >= 4 && D % 4 == 0 && H1 % 4 == 0 && H2 % 4 == 0 && H1 > 64 && H2 > 64 && OUT >= 1 && OUT <= 32 &&
           pick_variant(H1, H2, &nt1, &nt2) && nt1 > 2;
}

extern "C" int smx_mlp3_forward_rows_f32(const smx_mlp3_t* net, const float* x, int64_t rows, float* h1, float* h2,
                                         float* out, int32_t out_act, int32_t out_ld, float* packed,
                                         size_t packed_bytes, const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(net && x && h1 && h2 && out && packed, SMX_E_NULL);
    SMX_REQUIRE(rows > 0, SMX_E_SHAPE);
    if (!smx_mlp3_forward_rows_supported(net->D, net->H1, net->H2, net->OUT)) return SMX_E_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)h1 | (uintptr_t)h2) & 15) != 0) return SMX_E_UNSUPPORTED;
    int rc = smx_mlp3_pack_f32(net, packed, packed_bytes, stream);
    if (rc) return rc;
    FusedArgs A;
    A.packed = packed;
    A.x_main = x; A.x_tail = x;
    A.zmean = nullptr; A.zstd = nullptr;
    A.out = out;
    A.total_rows = (long)rows;
    A.T0 = 1; A.T1 = 0;
    A.D = net->D; A.OUT = net->OUT; A.KC1 = (net->D + 31) / 32;
    A.out_act = out_act;
    A.tbuf = nullptr; A.exp = 0; A.xvec = 1;
    A.h1_out = h1; A.h2_out = h2;
    A.H1 = net->H1; A.H2 = net->H2; A.out_ld = out_ld > 0 ? out_ld : net->OUT;
    A.stop = (const int*)stop_flag;
    return smx_rows16_launch(A, net->H1, net->H2, smx_s(stream));
}
