// Fused z-filter + Linear-ReLU-Linear-ReLU-Linear[-Tanh] forward, second generation: the same pass as
// smx_mlp3_fused.hip (PPOLearner._gae_and_return's critic over every step, surreal/learner/ppo.py:376-386
// -> surreal/model/ppo_net.py:284-315 -> model_builders/builders.py:159-175) on 16-row wavefronts.
//
// What changed against the 32-row kernel, and why (profiles/r02_pmc_mfma.json: MFMA busy 74 %, and
// 11 % of the issued MFMA work was padding):
//   * v_mfma_f32_16x16x4_f32 instead of 32x32x2: feature tiles of 16, so 300 -> 304 and 200 -> 208
//     features are issued instead of 320 / 224 (2.7 % padding instead of 11 %).  Same FLOP rate per
//     cycle, twice the LDS operand reads per FLOP -- LDS sits at a quarter of its bandwidth.
//   * one wavefront owns 16 data rows end to end; its accumulators are 19 x 4 + 13 x 4 = 128
//     registers instead of 272, so a workgroup is EIGHT wavefronts (128 rows) and every SIMD holds
//     two: while one waits for its LDS fragments, its staging stores or the hand-over between layers,
//     the other one issues MFMAs.
//   * the C/D fragment of 16x16x4 holds, per lane, data row (lane & 15) and features 4 (lane >> 4) + r;
//     with the K index of a 16-wide sub-chunk taken as k = 4 (lane >> 4) + step, register r of tile t
//     IS the B operand of step r of sub-chunk t of the next layer: activations stay in registers.
//   * the packed weights (smx_mlp3_pack_f32: [K chunk of 32][feature rows padded to 32][32]) are read
//     as they are; rows past the last 16-feature tile are staged but never used.
#include "smx_mlp3_fused.inc.h"

namespace {

constexpr int ROWS16 = 128;     // 8 wavefronts x 16 data rows
constexpr int LDS16 = 36;       // floats per staged row (32 + 4 pad): conflict-free ds_read_b128
constexpr int NTHR = 512;

#ifdef SMX_FUSED_TIMING
#define TSTAMP(i) do { if (A.tbuf && threadIdx.x == 0) A.tbuf[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
#define TLOOP(i) do { if (c == 5) TSTAMP(8 + (i)); } while (0)
#ifdef SMX_FUSED_EXP
#define EXP(bit) (SMX_FUSED_EXP & (bit))      // compile-time ablations (scripts/bench_fused.py)
#else
#define EXP(bit) 0
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    const uintptr_t u = (uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void* q = (void*)(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ float4 ld16(rsrc_t R, unsigned off) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(R, off, 0, 0);
    return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
}

__device__ __forceinline__ float relu16(float v) { return (v < 0.f) ? 0.f : v; }

// (x - m) * (1 / s), NaN-preserving clip to [-5, 5] (z_filter.py:77-79), selected branch-free; the
// same arithmetic as the 32-row kernel's stage_x
__device__ __forceinline__ float zf16(float x, float m, float r) {
    float v = (x - m) * r;
    if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);
    return v;
}
__device__ __forceinline__ void stage_x16(float* dst, float4 v, const float4 zm, const float4 rz, const bool hasz,
                                          const bool ok) {
    const float a = zf16(v.x, zm.x, rz.x), b = zf16(v.y, zm.y, rz.y);
    const float c = zf16(v.z, zm.z, rz.z), d = zf16(v.w, zm.w, rz.w);
    v.x = ok ? (hasz ? a : v.x) : 0.f;
    v.y = ok ? (hasz ? b : v.y) : 0.f;
    v.z = ok ? (hasz ? c : v.z) : 0.f;
    v.w = ok ? (hasz ? d : v.w) : 0.f;
    *reinterpret_cast<float4*>(dst) = v;
}

// NT1 / NT2: 16-feature tiles of the two hidden layers (H1 <= 16 NT1, H2 <= 16 NT2)
// SAVE: the hidden activations h1 / h2 (after bias + ReLU) also go to memory, row-major, straight from the accumulators
// (a lane holds 4 consecutive features of one row per tile: one 16-byte store; the four lane groups of a row fill a
// 64-byte segment) -- the forward half of a training step over B*E ~ 10^5 rows (the MLPs on top of a stem)
template <int NT1, int NT2, bool OUT1, bool SAVE>
__global__ __launch_bounds__(NTHR, 1) void mlp3_rows16_kernel(FusedArgs A) {
    constexpr int P1 = (NT1 + 1) / 2, P2 = (NT2 + 1) / 2;      // 32-feature tiles of the packed layout
    constexpr int NW1 = (P1 * 32 + 63) / 64, NW2 = (P2 * 32 + 63) / 64;   // staging passes of 64 rows
    constexpr int WR = 64 * (NW1 > NW2 ? NW1 : NW2);
    static_assert(NW1 <= 5 && NW2 <= 4, "staging registers are written out for <= 5 / <= 4 passes");
    extern __shared__ float lds[];
    if (SAVE && A.stop && *A.stop) return;
    TSTAMP(0);
    float* Wb0 = lds;
    float* Wb1 = Wb0 + WR * LDS16;
    float* Xb0 = Wb1 + WR * LDS16;
    float* Xb1 = Xb0 + ROWS16 * LDS16;
    float* b1s = Xb1 + ROWS16 * LDS16;
    float* b2s = b1s + P1 * 32;
    float* w3s = b2s + P2 * 32;        // OUT1: row 0 of W3 (P2*32); else b3 (32)
    float* zms = w3s + P2 * 32;        // z-filter mean and 1 / std per input k (KC1 * 32 each): the divisions
    float* rzl = zms + A.KC1 * 32;     // are done once per workgroup, not once per chunk and thread

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fm = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS16;
    const PackLayout L = pack_layout(P1, P2, A.KC1);
    const float* W1p = A.packed + L.w1;
    const float* W2p = A.packed + L.w2;
    const float* W3p = A.packed + L.w3;
    const bool hasz = A.zmean != nullptr;
    const rsrc_t rzm = make_rsrc(A.zmean, hasz ? (unsigned)A.D * 4u : 0u);   // absent: every load returns 0
    const rsrc_t rzs = make_rsrc(A.zstd, hasz ? (unsigned)A.D * 4u : 0u);

    // ---- staging geometry: thread -> (row srow + 64 i, 4 consecutive k at sk4) ----
    const int srow = tid >> 3, sk8 = tid & 7, sk4 = sk8 * 4;
    const float* xp0;
    const float* xp1;
    bool xok0, xok1;
    {
        const int T = A.T0 + A.T1;
        auto rowptr = [&](long r, bool& ok) {
            ok = r < A.total_rows;
            const long rr = ok ? r : 0;
            const long gq = rr / T;
            const int tt = (int)(rr - gq * T);
            return (tt < A.T0) ? A.x_main + (gq * A.T0 + tt) * (long)A.D
                               : A.x_tail + (gq * A.T1 + (tt - A.T0)) * (long)A.D;
        };
        xp0 = rowptr(row0 + srow, xok0);
        xp1 = rowptr(row0 + srow + 64, xok1);
    }
    for (int i = tid; i < P1 * 32; i += NTHR) b1s[i] = A.packed[L.b1 + i];
    for (int i = tid; i < P2 * 32; i += NTHR) b2s[i] = A.packed[L.b2 + i];
    for (int k = tid; k < A.KC1 * 32; k += NTHR) {
        const bool in = hasz && k < A.D;
        zms[k] = in ? A.zmean[k] : 0.f;
        rzl[k] = in ? 1.0f / A.zstd[k] : 1.0f;
    }
    const float b3v = A.packed[L.b3];      // read here: at the end its latency would be exposed
    if (OUT1) {
        for (int i = tid; i < P2 * 32; i += NTHR) w3s[i] = W3p[(size_t)(i >> 5) * 1024 + (i & 31)];
    } else {
        if (tid < 32) w3s[tid] = A.packed[L.b3 + tid];
    }

#define SMX_WLD(src, i) (src)[(srow + 64 * (i)) * 8 + sk8]
#define SMX_WST(dst, i, v) *reinterpret_cast<float4*>((dst) + (srow + 64 * (i)) * LDS16 + sk4) = (v)

    // ======================= layer 1: acc1[t] = W1[tile t] . x^T ===========================
    f32x4 acc1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    {   // chunk 0
        const float4* wsrc = reinterpret_cast<const float4*>(W1p);
        const int kc = (sk4 < A.D) ? sk4 : A.D - 4;
        const float4 x0 = *reinterpret_cast<const float4*>(xp0 + kc);
        const float4 x1 = *reinterpret_cast<const float4*>(xp1 + kc);
        const float4 zm = ld16(rzm, (unsigned)kc * 4u);
        float4 zs = ld16(rzs, (unsigned)kc * 4u);
        const float4 w0 = SMX_WLD(wsrc, 0), w1 = SMX_WLD(wsrc, 1), w2 = SMX_WLD(wsrc, 2), w3 = SMX_WLD(wsrc, 3);
        float4 w4 = w3;
        if (NW1 > 4) w4 = SMX_WLD(wsrc, 4);
        const float4 rz = make_float4(1.0f / zs.x, 1.0f / zs.y, 1.0f / zs.z, 1.0f / zs.w);
        const bool kin = sk4 < A.D;
        stage_x16(Xb0 + srow * LDS16 + sk4, x0, zm, rz, hasz, xok0 && kin);
        stage_x16(Xb0 + (srow + 64) * LDS16 + sk4, x1, zm, rz, hasz, xok1 && kin);
        SMX_WST(Wb0, 0, w0); SMX_WST(Wb0, 1, w1); SMX_WST(Wb0, 2, w2); SMX_WST(Wb0, 3, w3);
        if (NW1 > 4) SMX_WST(Wb0, 4, w4);
    }
    __syncthreads();
    TSTAMP(1);

    // Fragment reads run ONE HALF-PHASE AHEAD of the MFMAs that use them.  The eight wavefronts of a workgroup
    // leave every barrier together, so "read 20 fragments, then 76 MFMAs" keeps them in lock-step: all of them
    // read (LDS busy, matrix pipes idle, ~1400 cycles per sub-chunk measured), then all of them multiply.
    // The feature tiles are therefore split in two halves; while the MFMAs of one half issue, the fragments of
    // the next half (of this or the next sub-chunk) are already on their way.  The one barrier per chunk sits
    // after the LAST fragment read of the current buffers and after the staging stores of the next ones, so the
    // first fragments of the next chunk are prefetched under the final MFMA half-phase as well.
    // The staging work of the NEXT chunk (z-filter arithmetic, LDS stores) is cut into pieces that sit BETWEEN the
    // four k-step groups of a half-phase: a wavefront that issues one piece (~100 cycles of vector ALU / LDS) leaves
    // the matrix pipe to its SIMD partner instead of both of them doing all of it in front of the barrier
    // (measured: ~2400 cycles per chunk with the matrix pipes idle).
    // Inside a half-phase nothing is issued in a burst either: the fragment reads, the staging stores and the
    // global loads are cut into thirds that sit BETWEEN the four k-step groups (a workgroup's wavefronts leave the
    // barrier together; a burst of ~10 LDS reads or 7 global loads per wavefront keeps each of them ~500 cycles in
    // the LDS / texture-address queues with its MFMAs stuck behind in program order).
    constexpr int HA = (NT1 + 1) / 2, HB = NT1 - HA;     // tiles [0, HA) and [HA, NT1)
    constexpr int A1 = HA / 3, A2 = (2 * HA) / 3, B1 = HB / 3, B2 = (2 * HB) / 3;
    float4 fa[HA], fb[HB], bcur, bnxt;
#define SMX_W1(buf) ((buf) + woff)
#define SMX_RD_A(wrow, h, i0, i1)                                                 \
    _Pragma("unroll") for (int t = (i0); t < (i1); ++t)                           \
        fa[t] = *reinterpret_cast<const float4*>((wrow) + t * 16 * LDS16 + 16 * (h));
#define SMX_RD_B(wrow, h, i0, i1)                                                 \
    _Pragma("unroll") for (int t = (i0); t < (i1); ++t)                           \
        fb[t] = *reinterpret_cast<const float4*>((wrow) + (HA + t) * 16 * LDS16 + 16 * (h));
#define SMX_STEP_A(e)                                                             \
    _Pragma("unroll") for (int t = 0; t < HA; ++t) acc1[t] = MFMA16(fa[t].e, bq.e, acc1[t]);
#define SMX_STEP_B(e)                                                             \
    _Pragma("unroll") for (int t = 0; t < HB; ++t) acc1[HA + t] = MFMA16(fb[t].e, bq.e, acc1[HA + t]);
#define SMX_PIN __builtin_amdgcn_sched_barrier(0)
    const int xoff = (wv * 16 + fm) * LDS16 + 4 * g, woff = fm * LDS16 + 4 * g;

    // staging registers of the chunk after the next barrier (named scalars: see smx_mlp3_fused.hip)
    float4 x0, x1, w0, w1, w2, w3, w4;
    const int last = A.KC1 - 1;
    {
        const int cn = 1 < last ? 1 : last;
        const float4* wsrc = reinterpret_cast<const float4*>(W1p + (size_t)cn * P1 * 1024);
        const int k0n = 32 * cn + sk4;
        const int kc = (k0n < A.D) ? k0n : A.D - 4;          // unconditional, clamped (zeroed at the LDS store)
        // pinned in the loop's issue order (w0 .. w4, x0, x1): s_waitcnt vmcnt counts back from the newest load,
        // and where the two paths into the loop disagree about the order hipcc waits for vmcnt(0) -- which made
        // every weight store wait for the x rows from HBM
        SMX_PIN; w0 = SMX_WLD(wsrc, 0); SMX_PIN; w1 = SMX_WLD(wsrc, 1); SMX_PIN; w2 = SMX_WLD(wsrc, 2); SMX_PIN;
        w3 = SMX_WLD(wsrc, 3); SMX_PIN;
        w4 = w3;
        if (NW1 > 4) w4 = SMX_WLD(wsrc, 4);
        SMX_PIN;
        x0 = *reinterpret_cast<const float4*>(xp0 + kc); SMX_PIN;
        x1 = *reinterpret_cast<const float4*>(xp1 + kc); SMX_PIN;
    }
    bcur = *reinterpret_cast<const float4*>(Xb0 + xoff);
    SMX_RD_A(SMX_W1(Wb0), 0, 0, HA)
#ifdef SMX_FUSED_PRIO
    // A/B switch (scripts/build_variant_lib.py): static priority for the second-dispatched half of the workgroup, which
    // loses every VALU arbitration against its SIMD partner (MI355X_MICROARCH.md, "Two waves per SIMD", item 4)
    if (wv >= 4) __builtin_amdgcn_s_setprio(SMX_FUSED_PRIO);
#endif

    for (int c = 0; c < A.KC1; ++c) {
        const float* Wc = (c & 1) ? Wb1 : Wb0;
        const float* Xc = (c & 1) ? Xb1 : Xb0;
        float* Wn = (c & 1) ? Wb0 : Wb1;
        float* Xn = (c & 1) ? Xb0 : Xb1;
        // branch-free: the last iterations re-stage the last chunk (never used)
        const int cn = (c + 1 < A.KC1) ? c + 1 : c;            // the chunk the staging registers hold
        const int c2 = (c + 2 < A.KC1) ? c + 2 : last;         // the chunk whose loads are issued behind the barrier
        const bool kin = 32 * cn + sk4 < A.D;
        const int kc2 = (32 * c2 + sk4 < A.D) ? 32 * c2 + sk4 : A.D - 4;     // unconditional, clamped loads
        SMX_PIN;
        TLOOP(0);
        // fragment thirds are issued in front of k-step groups x, y, z of the half-phase BEFORE the one that uses
        // them (two groups of slack at the least); the staging pieces follow groups x, y, z
        {   // ---- sub-chunk 0, first half ----
            const float4 bq = bcur;
            SMX_RD_B(SMX_W1(Wc), 0, 0, B1) SMX_PIN;
            SMX_STEP_A(x) SMX_PIN; SMX_RD_B(SMX_W1(Wc), 0, B1, B2) SMX_PIN;
            SMX_STEP_A(y) SMX_PIN; SMX_RD_B(SMX_W1(Wc), 0, B2, HB) SMX_PIN;
            SMX_STEP_A(z) SMX_PIN;
            SMX_STEP_A(w) SMX_PIN;
        }
        TLOOP(1);
        // staging registers are re-loaded (chunk c + 2) right behind the store that frees them: a full iteration
        // (~5 us) ahead of their next use
        const float4* wsrc = reinterpret_cast<const float4*>(W1p + (size_t)c2 * P1 * 1024);
        {   // ---- sub-chunk 0, second half; the next chunk's weight rows go to LDS ----
            const float4 bq = bcur;
            bnxt = *reinterpret_cast<const float4*>(Xc + xoff + 16);
            SMX_RD_A(SMX_W1(Wc), 1, 0, A1)
            SMX_PIN;
            SMX_STEP_B(x) SMX_PIN;
            SMX_RD_A(SMX_W1(Wc), 1, A1, A2)
            if (!EXP(2)) { SMX_WST(Wn, 0, w0); SMX_WST(Wn, 1, w1); }
            if (!EXP(1)) { w0 = SMX_WLD(wsrc, 0); w1 = SMX_WLD(wsrc, 1); }
            SMX_PIN;
            SMX_STEP_B(y) SMX_PIN;
            SMX_RD_A(SMX_W1(Wc), 1, A2, HA)
            if (!EXP(2)) { SMX_WST(Wn, 2, w2); SMX_WST(Wn, 3, w3); }
            if (!EXP(1)) { w2 = SMX_WLD(wsrc, 2); w3 = SMX_WLD(wsrc, 3); }
            SMX_PIN;
            SMX_STEP_B(z) SMX_PIN;
            if (NW1 > 4 && !EXP(2)) SMX_WST(Wn, 4, w4);
            if (NW1 > 4 && !EXP(1)) w4 = SMX_WLD(wsrc, 4);
            SMX_PIN;
            SMX_STEP_B(w) SMX_PIN;
        }
        TLOOP(2);
        {   // ---- sub-chunk 1, first half; the next chunk's x rows are z-filtered into LDS ----
            const float4 bq = bnxt;
            SMX_RD_B(SMX_W1(Wc), 1, 0, B1)
            const float4 zm = *reinterpret_cast<const float4*>(zms + 32 * cn + sk4);
            const float4 rz = *reinterpret_cast<const float4*>(rzl + 32 * cn + sk4);
            SMX_PIN;
            SMX_STEP_A(x) SMX_PIN;
            SMX_RD_B(SMX_W1(Wc), 1, B1, B2)
            SMX_PIN;
            SMX_STEP_A(y) SMX_PIN;
            SMX_RD_B(SMX_W1(Wc), 1, B2, HB)
            // the empty asm re-defines the loaded registers HERE: without it the (pure) z-filter arithmetic
            // is emitted right behind the loads and waits for HBM there
            asm volatile("" : "+v"(x0.x), "+v"(x0.y), "+v"(x0.z), "+v"(x0.w));
            if (!EXP(4)) stage_x16(Xn + srow * LDS16 + sk4, x0, zm, rz, hasz, xok0 && kin);
            if (!EXP(1)) x0 = *reinterpret_cast<const float4*>(xp0 + kc2);
            SMX_PIN;
            SMX_STEP_A(z) SMX_PIN;
            asm volatile("" : "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x1.w));
            if (!EXP(4)) stage_x16(Xn + (srow + 64) * LDS16 + sk4, x1, zm, rz, hasz, xok1 && kin);
            if (!EXP(1)) x1 = *reinterpret_cast<const float4*>(xp1 + kc2);
            SMX_PIN;
            SMX_STEP_A(w) SMX_PIN;
        }
        TLOOP(3);
        TLOOP(4);
        if (!EXP(8)) __syncthreads();
        TLOOP(5);
        {   // ---- sub-chunk 1, second half; first fragments of the next chunk ----
            const float4 bq = bnxt;
            bcur = *reinterpret_cast<const float4*>(Xn + xoff);
            SMX_RD_A(SMX_W1(Wn), 0, 0, A1)
            SMX_PIN;
            SMX_STEP_B(x) SMX_PIN;
            SMX_RD_A(SMX_W1(Wn), 0, A1, A2)
            SMX_PIN;
            SMX_STEP_B(y) SMX_PIN;
            SMX_RD_A(SMX_W1(Wn), 0, A2, HA)
            SMX_PIN;
            SMX_STEP_B(z) SMX_PIN;
            SMX_STEP_B(w) SMX_PIN;
        }
        TLOOP(6);
    }
#undef SMX_STEP_A
#undef SMX_STEP_B
#undef SMX_RD_A
#undef SMX_RD_B
    __syncthreads();            // the trailing prefetch read a staging buffer that layer 2 overwrites

    TSTAMP(2);
    // bias + ReLU in the C-fragment layout: register r of tile t holds feature 16 t + 4 g + r
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const float4 bb = *reinterpret_cast<const float4*>(b1s + 16 * t + 4 * g);
        acc1[t][0] = relu16(acc1[t][0] + bb.x);
        acc1[t][1] = relu16(acc1[t][1] + bb.y);
        acc1[t][2] = relu16(acc1[t][2] + bb.z);
        acc1[t][3] = relu16(acc1[t][3] + bb.w);
    }
    if (SAVE && A.h1_out) {
        const long r = row0 + wv * 16 + fm;
        if (r < A.total_rows) {
            float* dst = A.h1_out + r * (long)A.H1 + 4 * g;
#pragma unroll
            for (int t = 0; t < NT1; ++t)
                if (16 * t + 4 * g < A.H1)
                    *reinterpret_cast<float4*>(dst + 16 * t) = make_float4(acc1[t][0], acc1[t][1], acc1[t][2], acc1[t][3]);
        }
    }

    // ======================= layer 2: acc2[u] = W2[tile u] . h1^T ==========================
    f32x4 acc2[NT2];
#pragma unroll
    for (int u = 0; u < NT2; ++u) acc2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 v0, v1, v2, v3;
#define SMX_LD2(p_)                                                                                        \
    do {                                                                                                   \
        const float4* wsrc = reinterpret_cast<const float4*>(W2p + (size_t)(p_) * P2 * 1024);             \
        v0 = SMX_WLD(wsrc, 0); v1 = SMX_WLD(wsrc, 1); v2 = SMX_WLD(wsrc, 2);                              \
        v3 = v2;                                                                                           \
        if (NW2 > 3) v3 = SMX_WLD(wsrc, 3);                                                               \
    } while (0)
    SMX_LD2(0);
    SMX_WST(Wb0, 0, v0); SMX_WST(Wb0, 1, v1); SMX_WST(Wb0, 2, v2);
    if (NW2 > 3) SMX_WST(Wb0, 3, v3);
    if (P1 > 1) SMX_LD2(1);
    __syncthreads();
    TSTAMP(3);
    // same scheme; the B operands are the layer-1 accumulators
    constexpr int UA = (NT2 + 1) / 2, UB = NT2 - UA;
    constexpr int C1 = UA / 3, C2 = (2 * UA) / 3, D1 = UB / 3, D2 = (2 * UB) / 3;
    float4 ga[UA], gb[UB];
#define SMX_RD2_A(wrow, h, i0, i1)                                                \
    _Pragma("unroll") for (int u = (i0); u < (i1); ++u)                           \
        ga[u] = *reinterpret_cast<const float4*>((wrow) + u * 16 * LDS16 + 16 * (h));
#define SMX_RD2_B(wrow, h, i0, i1)                                                \
    _Pragma("unroll") for (int u = (i0); u < (i1); ++u)                           \
        gb[u] = *reinterpret_cast<const float4*>((wrow) + (UA + u) * 16 * LDS16 + 16 * (h));
#define SMX_STEP2_A(e, r)                                                         \
    _Pragma("unroll") for (int u = 0; u < UA; ++u) acc2[u] = MFMA16(ga[u].e, bq[r], acc2[u]);
#define SMX_STEP2_B(e, r)                                                         \
    _Pragma("unroll") for (int u = 0; u < UB; ++u) acc2[UA + u] = MFMA16(gb[u].e, bq[r], acc2[UA + u]);
    SMX_RD2_A(Wb0 + woff, 0, 0, UA)
#pragma unroll
    for (int p = 0; p < P1; ++p) {             // K chunk p of layer 2 = h1 tiles 2p, 2p + 1
        const float* Wc = (p & 1) ? Wb1 : Wb0;
        float* Wn = (p & 1) ? Wb0 : Wb1;
        const bool lastp = (p + 1 == P1);
        const bool odd_tail = (2 * p + 1 >= NT1);          // compile-time after unrolling: tile 2p+1 does not exist
        SMX_PIN;
        {
            const f32x4 bq = acc1[2 * p];
            SMX_STEP2_A(x, 0) SMX_PIN; SMX_RD2_B(Wc + woff, 0, 0, D1) SMX_PIN;
            SMX_STEP2_A(y, 1) SMX_PIN; SMX_RD2_B(Wc + woff, 0, D1, D2) SMX_PIN;
            SMX_STEP2_A(z, 2) SMX_PIN; SMX_RD2_B(Wc + woff, 0, D2, UB) SMX_PIN;
            SMX_STEP2_A(w, 3) SMX_PIN;
        }
        {
            const f32x4 bq = acc1[2 * p];
            SMX_STEP2_B(x, 0) SMX_PIN;
            if (!odd_tail) { SMX_RD2_A(Wc + woff, 1, 0, C1) }
            if (!lastp) { SMX_WST(Wn, 0, v0); SMX_WST(Wn, 1, v1); }
            SMX_PIN;
            SMX_STEP2_B(y, 1) SMX_PIN;
            if (!odd_tail) { SMX_RD2_A(Wc + woff, 1, C1, C2) }
            if (!lastp) { SMX_WST(Wn, 2, v2); if (NW2 > 3) SMX_WST(Wn, 3, v3); }
            SMX_PIN;
            SMX_STEP2_B(z, 2) SMX_PIN;
            if (!odd_tail) { SMX_RD2_A(Wc + woff, 1, C2, UA) }
            SMX_PIN;
            SMX_STEP2_B(w, 3) SMX_PIN;
        }
        if (!odd_tail) {
            const f32x4 bq = acc1[2 * p + 1 < NT1 ? 2 * p + 1 : 0];
            SMX_STEP2_A(x, 0) SMX_PIN; SMX_RD2_B(Wc + woff, 1, 0, D1) SMX_PIN;
            SMX_STEP2_A(y, 1) SMX_PIN; SMX_RD2_B(Wc + woff, 1, D1, D2) SMX_PIN;
            SMX_STEP2_A(z, 2) SMX_PIN; SMX_RD2_B(Wc + woff, 1, D2, UB) SMX_PIN;
            SMX_STEP2_A(w, 3) SMX_PIN;
        }
        if (!lastp) __syncthreads();
        if (!odd_tail) {
            const f32x4 bq = acc1[2 * p + 1 < NT1 ? 2 * p + 1 : 0];
            SMX_STEP2_B(x, 0) SMX_PIN;
            if (!lastp) { SMX_RD2_A(Wn + woff, 0, 0, C1) }
            if (p + 2 < P1) SMX_LD2(p + 2);
            SMX_PIN;
            SMX_STEP2_B(y, 1) SMX_PIN;
            if (!lastp) { SMX_RD2_A(Wn + woff, 0, C1, C2) }
            SMX_PIN;
            SMX_STEP2_B(z, 2) SMX_PIN;
            if (!lastp) { SMX_RD2_A(Wn + woff, 0, C2, UA) }
            SMX_PIN;
            SMX_STEP2_B(w, 3) SMX_PIN;
        } else if (!lastp) {
            SMX_RD2_A(Wn + woff, 0, 0, UA)
            if (p + 2 < P1) SMX_LD2(p + 2);
        }
    }
#undef SMX_RD2_A
#undef SMX_RD2_B
#undef SMX_STEP2_A
#undef SMX_STEP2_B
#undef SMX_LD2
    TSTAMP(4);
#pragma unroll
    for (int u = 0; u < NT2; ++u) {
        const float4 bb = *reinterpret_cast<const float4*>(b2s + 16 * u + 4 * g);
        acc2[u][0] = relu16(acc2[u][0] + bb.x);
        acc2[u][1] = relu16(acc2[u][1] + bb.y);
        acc2[u][2] = relu16(acc2[u][2] + bb.z);
        acc2[u][3] = relu16(acc2[u][3] + bb.w);
    }
    if (SAVE && A.h2_out) {
        const long r = row0 + wv * 16 + fm;
        if (r < A.total_rows) {
            float* dst = A.h2_out + r * (long)A.H2 + 4 * g;
#pragma unroll
            for (int u = 0; u < NT2; ++u)
                if (16 * u + 4 * g < A.H2)
                    *reinterpret_cast<float4*>(dst + 16 * u) = make_float4(acc2[u][0], acc2[u][1], acc2[u][2], acc2[u][3]);
        }
    }

    // ======================= layer 3 ========================================================
    const long myrow = row0 + wv * 16 + fm;
    if (OUT1) {
        // one output: a 4 NT2-term dot product per lane, summed over the four lane groups
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            const float4 ww = *reinterpret_cast<const float4*>(w3s + 16 * u + 4 * g);
            v = fmaf(acc2[u][0], ww.x, v);
            v = fmaf(acc2[u][1], ww.y, v);
            v = fmaf(acc2[u][2], ww.z, v);
            v = fmaf(acc2[u][3], ww.w, v);
        }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        v += b3v;
        if (A.out_act == SMX_ACT_TANH) v = tanhf(v);
        if (g == 0 && myrow < A.total_rows) A.out[SAVE ? myrow * A.out_ld : myrow] = v;
    } else {
        // OUT <= 32 outputs: one or two more MFMA tiles; all of W3 (P2 chunks of 32 rows) fits one staging buffer
        f32x4 acc3 = f32x4{0.f, 0.f, 0.f, 0.f}, acc3b = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float4* src = reinterpret_cast<const float4*>(W3p);
            for (int i = tid; i < P2 * 32 * 8; i += NTHR)
                *reinterpret_cast<float4*>(Wb0 + (i >> 3) * LDS16 + 4 * (i & 7)) = src[i];
        }
        __syncthreads();
        const float* wrow = Wb0 + fm * LDS16 + 4 * g;
        const bool two = A.OUT > 16;             // workgroup-uniform
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(wrow + (u >> 1) * 32 * LDS16 + 16 * (u & 1));
            acc3 = MFMA16(a.x, acc2[u][0], acc3);
            acc3 = MFMA16(a.y, acc2[u][1], acc3);
            acc3 = MFMA16(a.z, acc2[u][2], acc3);
            acc3 = MFMA16(a.w, acc2[u][3], acc3);
        }
        if (two) {
#pragma unroll
            for (int u = 0; u < NT2; ++u) {
                const float4 a = *reinterpret_cast<const float4*>(wrow + ((u >> 1) * 32 + 16) * LDS16 + 16 * (u & 1));
                acc3b = MFMA16(a.x, acc2[u][0], acc3b);
                acc3b = MFMA16(a.y, acc2[u][1], acc3b);
                acc3b = MFMA16(a.z, acc2[u][2], acc3b);
                acc3b = MFMA16(a.w, acc2[u][3], acc3b);
            }
        }
        if (myrow < A.total_rows) {
            const int ld = SAVE ? A.out_ld : A.OUT;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int o = (r < 4) ? 4 * g + r : 16 + 4 * g + (r - 4);
                if (o < A.OUT) {
                    float v = ((r < 4) ? acc3[r & 3] : acc3b[r & 3]) + w3s[o];
                    if (A.out_act == SMX_ACT_TANH) v = tanhf(v);
                    else if (A.out_act == SMX_ACT_RELU) v = relu16(v);
                    A.out[myrow * ld + o] = v;
                }
            }
        }
    }
    TSTAMP(5);
#undef SMX_WLD
#undef SMX_WST
}

template <int NT1, int NT2, bool SAVE>
int launch16(const FusedArgs& A, hipStream_t st) {
    constexpr int P1 = (NT1 + 1) / 2, P2 = (NT2 + 1) / 2;
    constexpr int NW1 = (P1 * 32 + 63) / 64, NW2 = (P2 * 32 + 63) / 64;
    constexpr int WR = 64 * (NW1 > NW2 ? NW1 : NW2);
    const size_t lds = (size_t)(2 * WR * LDS16 + 2 * ROWS16 * LDS16 + P1 * 32 + 2 * P2 * 32 + 2 * A.KC1 * 32) * sizeof(float);
    if (lds > 160 * 1024) return SMX_E_UNSUPPORTED;     // very wide inputs: the 32-row kernel takes them
    const unsigned grid = (unsigned)((A.total_rows + ROWS16 - 1) / ROWS16);
    void (*k)(FusedArgs) = (A.OUT == 1) ? mlp3_rows16_kernel<NT1, NT2, true, SAVE> : mlp3_rows16_kernel<NT1, NT2, false, SAVE>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), lds, st, A);
    e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}

}  // namespace

int smx_rows16_launch(const FusedArgs& A, int H1, int H2, hipStream_t st) {
    // the packed layout must be the 10 / 7 (x 32 features) one of smx_mlp3_pack_f32's large variant
    const bool save = A.h1_out || A.h2_out;
    if (!A.xvec || A.D < 4 || A.OUT > 32 || H1 <= 64 || H2 <= 64 || H1 > 320 || H2 > 224) return SMX_E_UNSUPPORTED;
    if (save) {
        if ((H1 | H2) & 3) return SMX_E_UNSUPPORTED;
        if (H1 <= 304 && H2 <= 208) return launch16<19, 13, true>(A, st);
        return launch16<20, 14, true>(A, st);
    }
    if (H1 <= 304 && H2 <= 208) return launch16<19, 13, false>(A, st);
    return launch16<20, 14, false>(A, st);
}
