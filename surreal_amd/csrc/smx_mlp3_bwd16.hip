// Data gradients of the 3-layer MLP over MANY rows in ONE launch: dz3 -> dz2 -> dz1 -> dx, the mirror image of
// smx_mlp3_rows16.hip (loss.backward() through Linear-ReLU-Linear-ReLU-Linear on top of an LSTM stem over B*E ~ 10^5 rows:
// surreal/learner/ppo.py:227-353 -> surreal/model/ppo_net.py:284-315; the reference leaves it to autograd).
//
// The layered path runs three GEMM launches -- dz2 = (dz3 W3) * relu'(h2), dz1 = (dz2 W2) * relu'(h1), dx = dz1 W1 -- and
// moves every intermediate through memory between them: 196 + 258 + 111 us at 126 976 rows
// (profiles/r05_lstm_1024x128_kernel_stats_b.csv; 23 GFLOP: 41 TFLOP/s).  Here a wavefront owns 16 rows end to end, as in
// the forward kernel, with everything transposed (dz^T = W^T . dz_next^T: the MFMA N axis is the data row, the M axis the
// feature): the C fragment of one product -- per lane data row (lane & 15), features 4 (lane >> 4) + r of a tile -- IS the
// B operand of the next, so dz2 and dz1 go from the accumulators to memory once (the weight gradients need them) and on,
// in registers, into the next product.  The ReLU masks are read from the saved activations in the same fragment layout.
// Weights: transposed copies packed like the forward ones ([32-wide K chunk][feature rows][32]), refreshed per call
// (they change every epoch); streamed L2 -> registers -> LDS double-buffered, one barrier per chunk.
#include "smx_common.h"
#include "smx_mlp3_bwd16.h"

namespace {

constexpr int ROWS = 128;       // 8 wavefronts x 16 data rows
constexpr int LDW = 36;         // floats per staged row (32 + 4 pad): conflict-free ds_read_b128
constexpr int NTHR = 512;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// One product with the B operand in REGISTERS: out[t] (OT feature tiles) += W^T chunks . in (KT tiles of 16 k each).
// Wp: packed [PK = ceil(KT / 2) chunks][PO * 32 feature rows][32 k].  Wb0 / Wb1: LDS staging, PO * 32 rows of LDW floats.
// Every wavefront of the workgroup calls it (barriers inside).
// the staging registers of one weight chunk (<= 5 passes of 64 rows): requested with first_chunk() BEFORE the epilogue of
// the previous product (mask loads, stores) so that the chunk's L2 round trip is not exposed at the product's start
struct Chunk { float4 v0, v1, v2, v3, v4; };

template <int OT>
__device__ __forceinline__ void first_chunk(Chunk& C, const float* __restrict__ Wp) {
    constexpr int PO = (OT + 1) / 2;
    constexpr int NW = (PO * 32 + 63) / 64;
    const int tid = threadIdx.x;
    const int srow = tid >> 3, sk8 = tid & 7;
    const float4* src = reinterpret_cast<const float4*>(Wp);
    C.v0 = C.v1 = C.v2 = C.v3 = C.v4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define B16_OK(i) ((i) < NW && (srow + 64 * (i) < PO * 32 || (PO * 32) % 64 == 0))
    if (B16_OK(0)) C.v0 = src[(srow + 0) * 8 + sk8];
    if (B16_OK(1)) C.v1 = src[(srow + 64) * 8 + sk8];
    if (B16_OK(2)) C.v2 = src[(srow + 128) * 8 + sk8];
    if (B16_OK(3)) C.v3 = src[(srow + 192) * 8 + sk8];
    if (B16_OK(4)) C.v4 = src[(srow + 256) * 8 + sk8];
#undef B16_OK
}

template <int KT, int OT>
__device__ __forceinline__ void product_from_registers(const f32x4 (&in)[KT], f32x4 (&out)[OT], const float* __restrict__ Wp,
                                                       float* Wb0, float* Wb1, const Chunk& first) {
    constexpr int PK = (KT + 1) / 2, PO = (OT + 1) / 2;
    constexpr int NW = (PO * 32 + 63) / 64;                // staging passes of 64 rows (512 threads x 16 bytes)
    static_assert(NW <= 5, "staging registers are written out for <= 5 passes");
    const int tid = threadIdx.x, lane = tid & 63;
    const int fm = lane & 15, g = lane >> 4;
    const int srow = tid >> 3, sk8 = tid & 7, sk4 = sk8 * 4;
    const int woff = fm * LDW + 4 * g;
    float4 v0 = first.v0, v1 = first.v1, v2 = first.v2, v3 = first.v3, v4 = first.v4;
#define B16_ROW_OK(i) ((i) < NW && (srow + 64 * (i) < PO * 32 || (PO * 32) % 64 == 0))
#define B16_LD(c)                                                                                              \
    do {                                                                                                       \
        const float4* src_ = reinterpret_cast<const float4*>(Wp + (size_t)(c) * PO * 1024);                   \
        if (B16_ROW_OK(0)) v0 = src_[(srow + 0) * 8 + sk8];                                                    \
        if (B16_ROW_OK(1)) v1 = src_[(srow + 64) * 8 + sk8];                                                   \
        if (B16_ROW_OK(2)) v2 = src_[(srow + 128) * 8 + sk8];                                                  \
        if (B16_ROW_OK(3)) v3 = src_[(srow + 192) * 8 + sk8];                                                  \
        if (B16_ROW_OK(4)) v4 = src_[(srow + 256) * 8 + sk8];                                                  \
    } while (0)
#define B16_ST(dst)                                                                                            \
    do {                                                                                                       \
        if (B16_ROW_OK(0)) *reinterpret_cast<float4*>((dst) + (srow + 0) * LDW + sk4) = v0;                    \
        if (B16_ROW_OK(1)) *reinterpret_cast<float4*>((dst) + (srow + 64) * LDW + sk4) = v1;                   \
        if (B16_ROW_OK(2)) *reinterpret_cast<float4*>((dst) + (srow + 128) * LDW + sk4) = v2;                  \
        if (B16_ROW_OK(3)) *reinterpret_cast<float4*>((dst) + (srow + 192) * LDW + sk4) = v3;                  \
        if (B16_ROW_OK(4)) *reinterpret_cast<float4*>((dst) + (srow + 256) * LDW + sk4) = v4;                  \
    } while (0)
#pragma unroll
    for (int t = 0; t < OT; ++t) out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                    // (the previous product's last fragment reads of Wb0 are done)
    B16_ST(Wb0);
    if (PK > 1) B16_LD(1);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PK; ++p) {
        const float* Wc = (p & 1) ? Wb1 : Wb0;
        float* Wn = (p & 1) ? Wb0 : Wb1;
        if (p + 1 < PK) {
            B16_ST(Wn);                 // chunk p + 1 (its readers of two chunks ago passed the last barrier)
            if (p + 2 < PK) B16_LD(p + 2);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (2 * p + h < KT) {
                const f32x4 bq = in[2 * p + h < KT ? 2 * p + h : 0];
#pragma unroll
                for (int t = 0; t < OT; ++t) {
                    const float4 a = *reinterpret_cast<const float4*>(Wc + woff + t * 16 * LDW + 16 * h);
                    out[t] = MFMA16(a.x, bq[0], out[t]);
                    out[t] = MFMA16(a.y, bq[1], out[t]);
                    out[t] = MFMA16(a.z, bq[2], out[t]);
                    out[t] = MFMA16(a.w, bq[3], out[t]);
                }
            }
        }
        if (p + 1 < PK) __syncthreads();
    }
#undef B16_LD
#undef B16_ST
#undef B16_ROW_OK
}

// The same product on the forward kernel's schedule (smx_mlp3_rows16.hip, layer 2): the eight wavefronts leave every barrier
// together, so "read all fragments, then all MFMAs" keeps them in lock-step -- LDS busy, matrix pipes idle, then the
// reverse.  The output tiles are split in two halves; while the MFMAs of one half issue, the fragments of the next half are
// on their way (in thirds, between the four k-step groups), the LDS stores of the next chunk sit between k-step groups too,
// and the one barrier per chunk comes after the last fragment read of the current buffer, so the first fragments of the next
// chunk are fetched under the chunk's last half-phase.  sched_barriers pin that order.
template <int KT, int OT>
__device__ __forceinline__ void product_interleaved(const f32x4 (&in)[KT], f32x4 (&out)[OT], const float* __restrict__ Wp,
                                                    float* Wb0, float* Wb1, const Chunk& first) {
    constexpr int PK = (KT + 1) / 2, PO = (OT + 1) / 2;
    constexpr int NW = (PO * 32 + 63) / 64;
    static_assert(NW <= 5, "staging registers are written out for <= 5 passes");
    constexpr int UA = (OT + 1) / 2, UB = OT - UA;
    constexpr int C1 = UA / 3, C2 = (2 * UA) / 3, D1 = UB / 3, D2 = (2 * UB) / 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int fm = lane & 15, g = lane >> 4;
    const int srow = tid >> 3, sk8 = tid & 7, sk4 = sk8 * 4;
    const int woff = fm * LDW + 4 * g;
    float4 v0 = first.v0, v1 = first.v1, v2 = first.v2, v3 = first.v3, v4 = first.v4;
    float4 ga[UA], gb[UB > 0 ? UB : 1];
#define I16_OK(i) ((i) < NW && (srow + 64 * (i) < PO * 32 || (PO * 32) % 64 == 0))
#define I16_LD(c)                                                                                              \
    do {                                                                                                       \
        const float4* src_ = reinterpret_cast<const float4*>(Wp + (size_t)(c) * PO * 1024);                   \
        if (I16_OK(0)) v0 = src_[(srow + 0) * 8 + sk8];                                                        \
        if (I16_OK(1)) v1 = src_[(srow + 64) * 8 + sk8];                                                       \
        if (I16_OK(2)) v2 = src_[(srow + 128) * 8 + sk8];                                                      \
        if (I16_OK(3)) v3 = src_[(srow + 192) * 8 + sk8];                                                      \
        if (I16_OK(4)) v4 = src_[(srow + 256) * 8 + sk8];                                                      \
    } while (0)
#define I16_ST(dst, i, v) do { if (I16_OK(i)) *reinterpret_cast<float4*>((dst) + (srow + 64 * (i)) * LDW + sk4) = (v); } while (0)
#define I16_RD_A(wrow, h, i0, i1)                                                 \
    _Pragma("unroll") for (int u = (i0); u < (i1); ++u)                           \
        ga[u] = *reinterpret_cast<const float4*>((wrow) + u * 16 * LDW + 16 * (h));
#define I16_RD_B(wrow, h, i0, i1)                                                 \
    _Pragma("unroll") for (int u = (i0); u < (i1); ++u)                           \
        gb[u] = *reinterpret_cast<const float4*>((wrow) + (UA + u) * 16 * LDW + 16 * (h));
#define I16_STEP_A(e, r)                                                          \
    _Pragma("unroll") for (int u = 0; u < UA; ++u) out[u] = MFMA16(ga[u].e, bq[r], out[u]);
#define I16_STEP_B(e, r)                                                          \
    _Pragma("unroll") for (int u = 0; u < UB; ++u) out[UA + u] = MFMA16(gb[u].e, bq[r], out[UA + u]);
#define I16_PIN __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int t = 0; t < OT; ++t) out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                    // (the previous product's last fragment reads of Wb0 are done)
    I16_ST(Wb0, 0, v0); I16_ST(Wb0, 1, v1); I16_ST(Wb0, 2, v2); I16_ST(Wb0, 3, v3); I16_ST(Wb0, 4, v4);
    if (PK > 1) I16_LD(1);
    __syncthreads();
    I16_RD_A(Wb0 + woff, 0, 0, UA)
#pragma unroll
    for (int p = 0; p < PK; ++p) {
        const float* Wc = (p & 1) ? Wb1 : Wb0;
        float* Wn = (p & 1) ? Wb0 : Wb1;
        const bool lastp = (p + 1 == PK);
        const bool odd_tail = (2 * p + 1 >= KT);            // compile-time after unrolling: tile 2 p + 1 does not exist
        I16_PIN;
        {
            const f32x4 bq = in[2 * p];
            I16_STEP_A(x, 0) I16_PIN; I16_RD_B(Wc + woff, 0, 0, D1) I16_PIN;
            I16_STEP_A(y, 1) I16_PIN; I16_RD_B(Wc + woff, 0, D1, D2) I16_PIN;
            I16_STEP_A(z, 2) I16_PIN; I16_RD_B(Wc + woff, 0, D2, UB) I16_PIN;
            I16_STEP_A(w, 3) I16_PIN;
        }
        {
            const f32x4 bq = in[2 * p];
            I16_STEP_B(x, 0) I16_PIN;
            if (!odd_tail) { I16_RD_A(Wc + woff, 1, 0, C1) }
            if (!lastp) { I16_ST(Wn, 0, v0); I16_ST(Wn, 1, v1); }
            I16_PIN;
            I16_STEP_B(y, 1) I16_PIN;
            if (!odd_tail) { I16_RD_A(Wc + woff, 1, C1, C2) }
            if (!lastp) { I16_ST(Wn, 2, v2); I16_ST(Wn, 3, v3); }
            I16_PIN;
            I16_STEP_B(z, 2) I16_PIN;
            if (!odd_tail) { I16_RD_A(Wc + woff, 1, C2, UA) }
            if (!lastp) { I16_ST(Wn, 4, v4); }
            I16_PIN;
            I16_STEP_B(w, 3) I16_PIN;
        }
        if (!odd_tail) {
            const f32x4 bq = in[2 * p + 1 < KT ? 2 * p + 1 : 0];
            I16_STEP_A(x, 0) I16_PIN; I16_RD_B(Wc + woff, 1, 0, D1) I16_PIN;
            I16_STEP_A(y, 1) I16_PIN; I16_RD_B(Wc + woff, 1, D1, D2) I16_PIN;
            I16_STEP_A(z, 2) I16_PIN; I16_RD_B(Wc + woff, 1, D2, UB) I16_PIN;
            I16_STEP_A(w, 3) I16_PIN;
        }
        if (!lastp) __syncthreads();
        if (!odd_tail) {
            const f32x4 bq = in[2 * p + 1 < KT ? 2 * p + 1 : 0];
            I16_STEP_B(x, 0) I16_PIN;
            if (!lastp) { I16_RD_A(Wn + woff, 0, 0, C1) }
            if (p + 2 < PK) I16_LD(p + 2);
            I16_PIN;
            I16_STEP_B(y, 1) I16_PIN;
            if (!lastp) { I16_RD_A(Wn + woff, 0, C1, C2) }
            I16_PIN;
            I16_STEP_B(z, 2) I16_PIN;
            if (!lastp) { I16_RD_A(Wn + woff, 0, C2, UA) }
            I16_PIN;
            I16_STEP_B(w, 3) I16_PIN;
        } else if (!lastp) {
            I16_RD_A(Wn + woff, 0, 0, UA)
            if (p + 2 < PK) I16_LD(p + 2);
        }
    }
#undef I16_OK
#undef I16_LD
#undef I16_ST
#undef I16_RD_A
#undef I16_RD_B
#undef I16_STEP_A
#undef I16_STEP_B
#undef I16_PIN
}

// out[t][r] *= (act[row, 16 t + 4 g + r] > 0), then the tile goes to dst[row, 16 t + 4 g ..] (16-byte stores)
template <int OT>
__device__ __forceinline__ void mask_and_store(f32x4 (&acc)[OT], const float* __restrict__ act, float* __restrict__ dst,
                                               const long row, const bool row_ok, const int F, const int g) {
    if (!row_ok) {
#pragma unroll
        for (int t = 0; t < OT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const float* ap = act ? act + row * (long)F + 4 * g : nullptr;
    float* dp = dst + row * (long)F + 4 * g;
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        if (16 * t + 4 * g < F) {
            if (ap) {
                const float4 m = *reinterpret_cast<const float4*>(ap + 16 * t);
                acc[t][0] = (m.x > 0.f) ? acc[t][0] : 0.f;
                acc[t][1] = (m.y > 0.f) ? acc[t][1] : 0.f;
                acc[t][2] = (m.z > 0.f) ? acc[t][2] : 0.f;
                acc[t][3] = (m.w > 0.f) ? acc[t][3] : 0.f;
            }
            *reinterpret_cast<float4*>(dp + 16 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        } else {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};          // features past F: zero K rows of the next product
        }
    }
}

// NT1 / NT2 / NTD: 16-feature tiles of H1, H2 and of the input width D
template <int NT1, int NT2, int NTD>
__global__ __launch_bounds__(NTHR, 1) void mlp3_bwd16_kernel(Bwd16Args A) {
    constexpr int P1 = (NT1 + 1) / 2, P2 = (NT2 + 1) / 2, PD = (NTD + 1) / 2;
    constexpr int WR = 32 * (P1 > P2 ? (P1 > PD ? P1 : PD) : (P2 > PD ? P2 : PD));
    extern __shared__ float lds[];
    if (A.stop && *A.stop) return;
    float* Wb0 = lds;
    float* Wb1 = Wb0 + WR * LDW;
    float* Xb = Wb1 + WR * LDW;                    // the dz3 tile: [ROWS][LDW], columns >= OUT zero
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fm = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS;
    const long myrow = row0 + wv * 16 + fm;
    const bool row_ok = myrow < A.rows;
    const int srow = tid >> 3, sk8 = tid & 7, sk4 = sk8 * 4;

    // ---- stage the dz3 tile and the (single) chunk of W3^T ------------------------------------------------------
    {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long r = row0 + srow + 64 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < A.rows) {
                const float* p = A.dz3 + r * (long)A.ld3;
                if (sk4 + 0 < A.OUT) v.x = p[sk4 + 0];
                if (sk4 + 1 < A.OUT) v.y = p[sk4 + 1];
                if (sk4 + 2 < A.OUT) v.z = p[sk4 + 2];
                if (sk4 + 3 < A.OUT) v.w = p[sk4 + 3];
            }
            *reinterpret_cast<float4*>(Xb + (srow + 64 * i) * LDW + sk4) = v;
        }
        const float4* src = reinterpret_cast<const float4*>(A.pt3);
        for (int i = tid; i < P2 * 32 * 8; i += NTHR)
            *reinterpret_cast<float4*>(Wb0 + (i >> 3) * LDW + 4 * (i & 7)) = src[i];
    }
    __syncthreads();

    // ======================= dz2^T = W3^T . dz3^T (K = OUT <= 32: one chunk) ================================
    f32x4 acc2[NT2];
    {
        const float* xr = Xb + (wv * 16 + fm) * LDW + 4 * g;
        const float4 b0 = *reinterpret_cast<const float4*>(xr);
        const float4 b1 = *reinterpret_cast<const float4*>(xr + 16);
        const bool two = A.OUT > 16;                   // workgroup-uniform
        const float* wr = Wb0 + fm * LDW + 4 * g;
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
            const float4 a = *reinterpret_cast<const float4*>(wr + u * 16 * LDW);
            c = MFMA16(a.x, b0.x, c); c = MFMA16(a.y, b0.y, c); c = MFMA16(a.z, b0.z, c); c = MFMA16(a.w, b0.w, c);
            if (two) {
                const float4 a1 = *reinterpret_cast<const float4*>(wr + u * 16 * LDW + 16);
                c = MFMA16(a1.x, b1.x, c); c = MFMA16(a1.y, b1.y, c); c = MFMA16(a1.z, b1.z, c); c = MFMA16(a1.w, b1.w, c);
            }
            acc2[u] = c;
        }
    }
    Chunk ck;
#ifndef SMX_BWD16_NOHOIST               // (A/B switch, scripts/build_variant_lib.py)
    first_chunk<NT1>(ck, A.pt2);        // (in flight under the mask loads and the dz2 stores)
#endif
    mask_and_store<NT2>(acc2, A.h2, A.dz2, myrow, row_ok, A.H2, g);
#ifdef SMX_BWD16_NOHOIST
    first_chunk<NT1>(ck, A.pt2);
#endif

    // ======================= dz1^T = W2^T . dz2^T ==============================================================
    f32x4 acc1[NT1];
#ifdef SMX_BWD16_SIMPLE                 // (A/B switch: the plain chunk loop)
    product_from_registers<NT2, NT1>(acc2, acc1, A.pt2, Wb0, Wb1, ck);
#else
    product_interleaved<NT2, NT1>(acc2, acc1, A.pt2, Wb0, Wb1, ck);
#endif
#ifndef SMX_BWD16_NOHOIST
    if (A.dx) first_chunk<NTD>(ck, A.pt1);
#endif
    mask_and_store<NT1>(acc1, A.h1, A.dz1, myrow, row_ok, A.H1, g);
#ifdef SMX_BWD16_NOHOIST
    if (A.dx) first_chunk<NTD>(ck, A.pt1);
#endif

    // ======================= dx^T = W1^T . dz1^T ================================================================
    if (A.dx) {
        f32x4 accx[NTD];
#ifdef SMX_BWD16_SIMPLE
        product_from_registers<NT1, NTD>(acc1, accx, A.pt1, Wb0, Wb1, ck);
#else
        product_interleaved<NT1, NTD>(acc1, accx, A.pt1, Wb0, Wb1, ck);
#endif
        mask_and_store<NTD>(accx, nullptr, A.dx, myrow, row_ok, A.D, g);
    }
}

// transposed packing: dst[c][f][k] = W[32 c + k][f] for a weight W [K rows, F columns] row-major (zero padded)
__global__ __launch_bounds__(256) void bwd16_pack_kernel(smx_mlp3_t net, float* __restrict__ pt, int P1, int P2, int PD) {
    const size_t n3 = (size_t)P2 * 1024, n2 = (size_t)P2 * P1 * 1024, n1 = (size_t)P1 * PD * 1024;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n3 + n2 + n1; i += (size_t)gridDim.x * 256) {
        float v = 0.f;
        if (i < n3) {                                   // W3 [OUT, H2]: K = OUT (one chunk), features H2
            const int k = (int)(i & 31), f = (int)(i >> 5);
            if (k < net.OUT && f < net.H2) v = net.W3[(size_t)k * net.H2 + f];
        } else if (i < n3 + n2) {                       // W2 [H2, H1]: K = H2, features H1
            const size_t e = i - n3;
            const int kk = (int)(e & 31);
            const size_t rc = e >> 5;
            const int f = (int)(rc % (size_t)(P1 * 32)), c = (int)(rc / (size_t)(P1 * 32));
            const int k = 32 * c + kk;
            if (k < net.H2 && f < net.H1) v = net.W2[(size_t)k * net.H1 + f];
        } else {                                        // W1 [H1, D]: K = H1, features D
            const size_t e = i - n3 - n2;
            const int kk = (int)(e & 31);
            const size_t rc = e >> 5;
            const int f = (int)(rc % (size_t)(PD * 32)), c = (int)(rc / (size_t)(PD * 32));
            const int k = 32 * c + kk;
            if (k < net.H1 && f < net.D) v = net.W1[(size_t)k * net.D + f];
        }
        pt[i] = v;
    }
}

template <int NT1, int NT2, int NTD>
int launch_bwd16(const smx_mlp3_t* net, Bwd16Args A, float* pt, hipStream_t st) {
    constexpr int P1 = (NT1 + 1) / 2, P2 = (NT2 + 1) / 2, PD = (NTD + 1) / 2;
    constexpr int WR = 32 * (P1 > P2 ? (P1 > PD ? P1 : PD) : (P2 > PD ? P2 : PD));
    const size_t n3 = (size_t)P2 * 1024, n2 = (size_t)P2 * P1 * 1024, n1 = (size_t)P1 * PD * 1024;
    unsigned pb = (unsigned)((n3 + n2 + n1 + 255) / 256);
    if (pb > 1024) pb = 1024;
    hipLaunchKernelGGL(bwd16_pack_kernel, dim3(pb), dim3(256), 0, st, *net, pt, P1, P2, PD);
    A.pt3 = pt; A.pt2 = pt + n3; A.pt1 = pt + n3 + n2;
    const size_t lds = (size_t)(2 * WR * LDW + ROWS * LDW) * sizeof(float);
    void (*k)(Bwd16Args) = mlp3_bwd16_kernel<NT1, NT2, NTD>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3((unsigned)((A.rows + ROWS - 1) / ROWS)), dim3(NTHR), lds, st, A);
    e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}

inline bool shape_ok(int D, int H1, int H2, int OUT) {
    return D >= 4 && D % 4 == 0 && D <= 128 && H1 % 4 == 0 && H2 % 4 == 0 && H1 > 64 && H1 <= 320 && H2 > 64 && H2 <= 224 &&
           OUT >= 1 && OUT <= 32;
}

}  // namespace

extern "C" int32_t smx_mlp3_dgrad_rows_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    return shape_ok(D, H1, H2, OUT) ? 1 : 0;
}

extern "C" int64_t smx_mlp3_dgrad_rows_ws_floats(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    if (!shape_ok(D, H1, H2, OUT)) return 0;
    const int P1 = 10, P2 = 7, PD = 4;                 // the largest instantiation
    return (int64_t)P2 * 1024 + (int64_t)P2 * P1 * 1024 + (int64_t)P1 * PD * 1024;
}

int smx_mlp3_dgrad_rows_launch(const smx_mlp3_t* net, const float* h1, const float* h2, const float* dz3, int64_t rows,
                               float* dz2, float* dz1, float* dx, float* packedT, int64_t packedT_floats,
                               const int32_t* stop_flag, hipStream_t st) {
    if (!shape_ok(net->D, net->H1, net->H2, net->OUT)) return SMX_E_UNSUPPORTED;
    if (packedT_floats < smx_mlp3_dgrad_rows_ws_floats(net->D, net->H1, net->H2, net->OUT)) return SMX_E_WORKSPACE;
    if ((((uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)dz2 | (uintptr_t)dz1 | (uintptr_t)dx | (uintptr_t)packedT) & 15) != 0)
        return SMX_E_UNSUPPORTED;
    Bwd16Args A;
    A.pt1 = A.pt2 = A.pt3 = nullptr;
    A.dz3 = dz3; A.ld3 = net->OUT;
    A.h1 = h1; A.h2 = h2; A.dz2 = dz2; A.dz1 = dz1; A.dx = dx;
    A.rows = (long)rows; A.D = net->D; A.H1 = net->H1; A.H2 = net->H2; A.OUT = net->OUT;
    A.stop = (const int*)stop_flag;
    const bool small_h = net->H1 <= 304 && net->H2 <= 208;
    if (small_h && net->D <= 112) return launch_bwd16<19, 13, 7>(net, A, packedT, st);
    if (small_h) return launch_bwd16<19, 13, 8>(net, A, packedT, st);
    if (net->D <= 112) return launch_bwd16<20, 14, 7>(net, A, packedT, st);
    return launch_bwd16<20, 14, 8>(net, A, packedT, st);
}
