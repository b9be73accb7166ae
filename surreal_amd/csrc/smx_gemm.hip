// Small-batch dense layers on FP32 MFMA for the PPO/DDPG epoch loops
// (surreal/learner/ppo.py:227-353: forward_actor/forward_critic on B rows, loss.backward()).
//
// These GEMMs are tiny (B = 1024 rows, K <= 1024) and latency-bound, so the decomposition is
// chosen for span, not reuse: one workgroup per 32x32 output tile, its 4 wavefronts split K
// (each wave owns every 4th 32-wide K block), operands go HBM/L2 -> registers directly in MFMA
// fragment order (everything here is L2-resident), partial tiles are combined through LDS in a
// fixed order (deterministic), and the epilogue (bias / ReLU / tanh / ReLU-mask / bias-gradient /
// sum-of-squares for clip_grad_norm_) is fused.  Up to 6 problems are grouped into one launch.
#include "smx_common.h"
#include "smx_wgrad.h"
#include "smx_mlp3_bwd16.h"
#include <string.h>
#include <stdlib.h>

namespace {

#include "smx_epoch_pack.inc.h"
#include "smx_adam.inc.h"

struct GemmProb {
    const float* A;
    const float* B;
    const float* bias;   // [N] or null
    const float* mask;   // [M, ldc] or null : C *= (mask > 0)
    float* C;
    float* dbias;        // [M] or null : dbias[m] = sum_k A(m,k)   (bias gradient of a dW GEMM)
    float* sumsq;        // per-tile sum of squares of C (+dbias) or null
    float* CT;           // optional transposed copy: CT[n * ldct + m] = C[m, n]
    int ldct;
    int lda, ldb, ldc, M, N, K;
    int a_kc, b_kc, act;
    int tiles_m, tiles_n, tile_base, a_mode, b_mode;   // 0 kc-vec, 1 kc-scalar, 2 k-strided
    unsigned a_bytes, b_bytes;   // extent of each operand for the buffer descriptors (< 2 GiB)
    const int* stop;     // device flag: non-zero -> this problem is skipped
    // split-K (weight gradients over many rows): `splits` workgroups share an output tile, each
    // sums k in [s*k_chunk, (s+1)*k_chunk) into its own slice C + s*c_split (dbias + s*M)
    int splits, k_chunk;
    long c_split;
};

constexpr int MAX_PROBS = 9;

struct GemmBatch {
    GemmProb p[MAX_PROBS];
    int n;
};

// Workgroup -> problem descriptor, in TWO batches of scalar loads.  Left to itself the compiler
// fetches the fields of G.p[pi] one by one where they are first used -- a dozen dependent
// kernarg round trips (~0.2 us each) in front of the first operand load, a third of the ~6 us a
// short dependent launch costs.  The empty asm statements "use" every field at once, so all the
// loads are issued back to back and waited for once.
struct TileBases {
    int tb[MAX_PROBS];
    int n, grid;
};

// batch 1: every problem's first tile, the problem count and the grid size (static kernarg offsets)
__device__ __forceinline__ TileBases load_tile_bases(const GemmBatch& G) {
    static_assert(MAX_PROBS == 9, "the asm operand list below names every entry");
    TileBases t;
#pragma unroll
    for (int k = 0; k < MAX_PROBS; ++k) t.tb[k] = G.p[k].tile_base;
    t.n = G.n;
    t.grid = (int)gridDim.x;
    asm volatile("" :: "s"(t.tb[0]), "s"(t.tb[1]), "s"(t.tb[2]), "s"(t.tb[3]), "s"(t.tb[4]), "s"(t.tb[5]),
                 "s"(t.tb[6]), "s"(t.tb[7]), "s"(t.tb[8]), "s"(t.n), "s"(t.grid));
    return t;
}

// batch 2: the whole descriptor of the problem that owns tile `bid`
__device__ __forceinline__ GemmProb select_problem(const GemmBatch& G, const TileBases& t, int bid, int* pi_out = nullptr) {
    int pi = 0;
#pragma unroll
    for (int k = 1; k < MAX_PROBS; ++k) pi += (k < t.n && bid >= t.tb[k]) ? 1 : 0;   // tile_base ascends
    if (pi_out) *pi_out = pi;
    GemmProb P = G.p[pi];
    asm volatile("" :: "s"(P.A), "s"(P.B), "s"(P.bias), "s"(P.mask), "s"(P.C), "s"(P.dbias), "s"(P.sumsq),
                 "s"(P.CT), "s"(P.stop), "s"(P.c_split), "s"(P.ldct), "s"(P.lda), "s"(P.ldb), "s"(P.ldc),
                 "s"(P.M), "s"(P.N), "s"(P.K), "s"(P.a_kc), "s"(P.b_kc), "s"(P.act), "s"(P.tiles_m),
                 "s"(P.tiles_n), "s"(P.tile_base), "s"(P.a_mode), "s"(P.b_mode), "s"(P.a_bytes),
                 "s"(P.b_bytes), "s"(P.splits), "s"(P.k_chunk));      // 29 of the 30 operands an asm may have
    return P;
}

// the early-exit flag of a problem (the KL stop of the policy epochs): the load is issued here, in
// front of the operand loads, and consumed with stop_taken() behind the main loop -- a stopped
// problem does its arithmetic and discards it (rare), every other launch saves a memory round trip
__device__ __forceinline__ int stop_load(const GemmProb& P) {
    return P.stop ? __builtin_nontemporal_load(P.stop) : 0;
}
__device__ __forceinline__ bool stop_taken(int v) { return __builtin_amdgcn_readfirstlane(v) != 0; }

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Operand fragment for one 8-wide k group: lane (i, kh) holds X(row0+i, k0 + 4kh + r), r = 0..3.
// Operands are read with BUFFER loads: the hardware bounds check returns 0 for an offset at or past
// num_records, so rows >= nrows and k >= K are handled by pointing the lane's byte offset out of
// range BEFORE the load.  Nothing touches the loaded registers until the MFMA that consumes them,
// which lets hipcc leave the whole prefetch ring in flight (a select or a branch after the load
// makes it wait for the data right there).
// MODE 0: K-contiguous, one 16-byte load (ld % 4 == 0, K % 4 == 0, 16-byte aligned base);
// MODE 1: K-contiguous, four 4-byte loads; MODE 2: K-strided (X(r, k) = X[k*ld + r]).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;   // extents are < 2 GiB (checked on the host)

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    const uintptr_t u = (uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void* q = (void*)(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// rowbase: byte offset of (row, k = 0), or OOB for a row past the matrix
template <int MODE>
__device__ __forceinline__ unsigned row_base(int ld, int row, int nrows) {
    if (row >= nrows) return OOB;
    return MODE == 2 ? (unsigned)row * 4u : (unsigned)row * (unsigned)ld * 4u;
}

template <int MODE>
__device__ __forceinline__ float4 load_frag(rsrc_t R, unsigned rowbase, int ld, int k0, int K) {
    float4 v;
    if (MODE == 0) {
        const unsigned off = (k0 < K) ? rowbase + (unsigned)k0 * 4u : OOB;
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(R, off, 0, 0);
        v.x = __uint_as_float(w.x); v.y = __uint_as_float(w.y);
        v.z = __uint_as_float(w.z); v.w = __uint_as_float(w.w);
    } else {
        const unsigned step = MODE == 1 ? 4u : (unsigned)ld * 4u;
        const unsigned o0 = rowbase + (unsigned)k0 * step;
        v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R, (k0 + 0 < K) ? o0 : OOB, 0, 0));
        v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R, (k0 + 1 < K) ? o0 + step : OOB, 0, 0));
        v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R, (k0 + 2 < K) ? o0 + 2 * step : OOB, 0, 0));
        v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R, (k0 + 3 < K) ? o0 + 3 * step : OOB, 0, 0));
    }
    return v;
}

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == SMX_ACT_RELU) return (v < 0.f) ? 0.f : v;
    if (act == SMX_ACT_TANH) return tanhf(v);
    return v;
}

// the 8 operand fragments (4 k-groups x {A, B}) of one 32-wide K super-block
struct Frag8 {
    float4 a[4], b[4];
};

struct Operands {
    rsrc_t ra, rb;
    unsigned abase, bbase;
    int lda, ldb, K;
};

template <int AM, int BM>
__device__ __forceinline__ void load_sb(Frag8& f, const Operands& O, int kb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f.a[q] = load_frag<AM>(O.ra, O.abase, O.lda, kb + 8 * q, O.K);
        f.b[q] = load_frag<BM>(O.rb, O.bbase, O.ldb, kb + 8 * q, O.K);
    }
}

__device__ __forceinline__ void mma_sb(f32x16& acc, float& asum, const Frag8& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc = MFMA32(f.a[q].x, f.b[q].x, acc);
        acc = MFMA32(f.a[q].y, f.b[q].y, acc);
        acc = MFMA32(f.a[q].z, f.b[q].z, acc);
        acc = MFMA32(f.a[q].w, f.b[q].w, acc);
        asum += (f.a[q].x + f.a[q].y) + (f.a[q].z + f.a[q].w);
    }
}

// K is cut into 32-wide super-blocks; wave wv owns sb = wv, wv+4, ...  Three register buffers form
// a ring so that up to three super-blocks of operand loads (24 x 16 B per lane) are in flight
// ahead of the MFMAs: these GEMMs are latency-bound, not bandwidth-bound.  Loads past K are
// out-of-range buffer loads (no memory traffic); MFMAs past K are skipped (wave-uniform).
struct KRange {          // the K range this workgroup sums (the whole K unless split)
    const float* A;
    const float* B;
    unsigned a_bytes, b_bytes;
    int K;
};

// `wv` / NW: this wave takes super-blocks wv, wv + NW, ... (NW = 4: the waves of a workgroup split K;
// NW = 1: the wave walks all of K by itself)
template <int AM, int BM, int NW = 4>
__device__ __forceinline__ void mainloop(const GemmProb& P, const KRange& R, int m0, int n0, int i,
                                         int kh, int wv, f32x16& acc, float& asum) {
    const int nsb = (R.K + 31) >> 5;
    const int kofs = 4 * kh;
    Operands O;
    O.ra = make_rsrc(R.A, R.a_bytes);
    O.rb = make_rsrc(R.B, R.b_bytes);
    O.abase = row_base<AM>(P.lda, m0 + i, P.M);
    O.bbase = row_base<BM>(P.ldb, n0 + i, P.N);
    O.lda = P.lda; O.ldb = P.ldb; O.K = R.K;
    Frag8 f0, f1, f2;
    // the barriers pin the issue order f0, f1, f2 (vmcnt is in-order: the wait for f0 must not
    // cover f1 / f2)
    load_sb<AM, BM>(f0, O, (wv + 0 * NW) * 32 + kofs);
    __builtin_amdgcn_sched_barrier(0);
    load_sb<AM, BM>(f1, O, (wv + 1 * NW) * 32 + kofs);
    __builtin_amdgcn_sched_barrier(0);
    load_sb<AM, BM>(f2, O, (wv + 2 * NW) * 32 + kofs);
    for (int sb = wv; sb < nsb; sb += 3 * NW) {
        __builtin_amdgcn_sched_barrier(0);
        mma_sb(acc, asum, f0);
        __builtin_amdgcn_sched_barrier(0);
        load_sb<AM, BM>(f0, O, (sb + 3 * NW) * 32 + kofs);
        __builtin_amdgcn_sched_barrier(0);
        if (sb + NW < nsb) mma_sb(acc, asum, f1);
        __builtin_amdgcn_sched_barrier(0);
        load_sb<AM, BM>(f1, O, (sb + 4 * NW) * 32 + kofs);
        __builtin_amdgcn_sched_barrier(0);
        if (sb + 2 * NW < nsb) mma_sb(acc, asum, f2);
        __builtin_amdgcn_sched_barrier(0);
        load_sb<AM, BM>(f2, O, (sb + 5 * NW) * 32 + kofs);
    }
}

// Coalesced operand path (both operands K-contiguous, 16-byte aligned).  In fragment order a load
// instruction touches 32 different rows -- 32 cache lines, 32 bytes of each -- and the CU's address
// unit, not the MFMA pipe, bounds the loop (scripts/micro/fwd_loop.hip: 2500 vs 1280 cycles per chunk
// on the same pattern).  Here a wave fetches its 32 x 32 block of each operand as whole 128-byte row
// segments (lane -> row (l >> 3) + 8 j, bytes 16 (l & 7)), passes it through a wave-private LDS tile and
// reads the fragments back with ds_read_b128 (rows of 36 floats: conflict-free).  LDS operations of a
// wave complete in order, so the tile needs no barrier and no double buffering: the fragments of
// super-block s are in registers before the data of s + 1 overwrites the tile.
struct CoBlk {
    float4 a[4], b[4];
};

struct CoOperands {
    rsrc_t ra, rb;
    unsigned arow[4], brow[4];      // byte offset of (row, k = 0) per instruction, or OOB
    int K;
};

__device__ __forceinline__ void load_co(CoBlk& g, const CoOperands& O, int kb, int kseg) {
    const int k = kb + kseg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32x4 wa = __builtin_amdgcn_raw_buffer_load_b128(O.ra, (k < O.K) ? O.arow[j] + (unsigned)k * 4u : OOB, 0, 0);
        const u32x4 wb = __builtin_amdgcn_raw_buffer_load_b128(O.rb, (k < O.K) ? O.brow[j] + (unsigned)k * 4u : OOB, 0, 0);
        g.a[j] = make_float4(__uint_as_float(wa.x), __uint_as_float(wa.y), __uint_as_float(wa.z), __uint_as_float(wa.w));
        g.b[j] = make_float4(__uint_as_float(wb.x), __uint_as_float(wb.y), __uint_as_float(wb.z), __uint_as_float(wb.w));
    }
}

__device__ __forceinline__ void stage_co(const CoBlk& g, float* sA, float* sB, int wofs) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        *(float4*)(sA + wofs + j * 8 * 36) = g.a[j];
        *(float4*)(sB + wofs + j * 8 * 36) = g.b[j];
    }
}

__device__ __forceinline__ void frags_co(Frag8& f, const float* sA, const float* sB, int rofs) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f.a[q] = *(const float4*)(sA + rofs + 8 * q);
        f.b[q] = *(const float4*)(sB + rofs + 8 * q);
    }
}

__device__ __forceinline__ void mainloop_co(const GemmProb& P, const KRange& R, int m0, int n0, int lane,
                                            int wv, float* stage, f32x16& acc, float& asum) {
    const int nsb = (R.K + 31) >> 5;
    CoOperands O;
    O.ra = make_rsrc(R.A, R.a_bytes);
    O.rb = make_rsrc(R.B, R.b_bytes);
    O.K = R.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        O.arow[j] = row_base<0>(P.lda, m0 + (lane >> 3) + 8 * j, P.M);
        O.brow[j] = row_base<0>(P.ldb, n0 + (lane >> 3) + 8 * j, P.N);
    }
    const int kseg = 4 * (lane & 7);
    float* sA = stage + wv * 2 * 32 * 36;
    float* sB = sA + 32 * 36;
    const int wofs = (lane >> 3) * 36 + kseg;                    // where this lane's words of a block go
    const int rofs = (lane & 31) * 36 + 4 * (lane >> 5);         // fragment (i = lane & 31, kh = lane >> 5)
    CoBlk ga, gb;
    Frag8 f0, f1;
    load_co(ga, O, (wv + 0) * 32, kseg);
    __builtin_amdgcn_sched_barrier(0);
    load_co(gb, O, (wv + 4) * 32, kseg);
    __builtin_amdgcn_sched_barrier(0);
    stage_co(ga, sA, sB, wofs);
    frags_co(f0, sA, sB, rofs);
    __builtin_amdgcn_sched_barrier(0);
    load_co(ga, O, (wv + 8) * 32, kseg);
    // two super-blocks per trip; blocks past K are zeros (their MFMAs run: at most one idle block)
    for (int sb = wv; sb < nsb; sb += 8) {
        __builtin_amdgcn_sched_barrier(0);
        stage_co(gb, sA, sB, wofs);               // data of sb + 4 (in flight since the previous trip)
        frags_co(f1, sA, sB, rofs);
        __builtin_amdgcn_sched_barrier(0);
        load_co(gb, O, (sb + 12) * 32, kseg);
        __builtin_amdgcn_sched_barrier(0);
        mma_sb(acc, asum, f0);
        __builtin_amdgcn_sched_barrier(0);
        stage_co(ga, sA, sB, wofs);               // data of sb + 8
        frags_co(f0, sA, sB, rofs);
        __builtin_amdgcn_sched_barrier(0);
        load_co(ga, O, (sb + 16) * 32, kseg);
        __builtin_amdgcn_sched_barrier(0);
        mma_sb(acc, asum, f1);
    }
}

// ALL_VEC: every problem of the batch is mode (0, 0) -- the host picks this instantiation, which
// fits 3 workgroups per CU (<= 168 VGPRs); the general one carries the scalar-load variants.
template <bool ALL_VEC>
__device__ __forceinline__ void gemm32_body(const GemmBatch& G) {
    __shared__ float red[4][32 * 33];
    __shared__ float stage[ALL_VEC ? 4 * 2 * 32 * 36 : 4];      // wave-private operand tiles (coalesced path)
    __shared__ float dbr[8][32];
    __shared__ float sred[16];

    // XCD-aware order: consecutive workgroup ids land on the 8 XCDs round-robin, each with its own
    // L2.  Give XCD x the x-th contiguous eighth of the tile list (tiles are row-major in M, so
    // that is a band of A rows and all of B) instead of every 8th tile -- otherwise all eight L2s
    // pull every operand of the launch across the fabric.
    const TileBases TB = load_tile_bases(G);
    int bid = blockIdx.x;
    {
        const int total = TB.grid, q = total >> 3, r = total & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int pi = 0;
    const GemmProb P = select_problem(G, TB, bid, &pi);
    const int stopv = stop_load(P);
    int tile = bid - P.tile_base;
    int sp = 0;
    KRange R;
    R.A = P.A; R.B = P.B; R.a_bytes = P.a_bytes; R.b_bytes = P.b_bytes; R.K = P.K;
    if (P.splits > 1) {
        sp = tile % P.splits;
        tile /= P.splits;
        const int k0 = sp * P.k_chunk;
        R.K = min(P.k_chunk, P.K - k0);
        R.A = P.A + (P.a_kc ? (size_t)k0 : (size_t)k0 * P.lda);
        R.B = P.B + (P.b_kc ? (size_t)k0 : (size_t)k0 * P.ldb);
        R.a_bytes = 4u * (P.a_kc ? (unsigned)(P.M - 1) * P.lda + R.K : (unsigned)(R.K - 1) * P.lda + P.M);
        R.b_bytes = 4u * (P.b_kc ? (unsigned)(P.N - 1) * P.ldb + R.K : (unsigned)(R.K - 1) * P.ldb + P.N);
    }
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and provably so
    const int i = lane & 31, kh = lane >> 5;

    // the epilogue's own inputs (4 outputs per thread: bias, ReLU mask) are requested HERE, in front
    // of the operand loads, instead of behind the last barrier where their latency is exposed
    // (named scalars: an array live across the main loop's scheduling barriers goes to scratch)
    const int orow = tid >> 3, oc0 = (tid & 7) * 4;
    const int m = m0 + orow;
    float pb0 = 0.f, pb1 = 0.f, pb2 = 0.f, pb3 = 0.f, pm0 = 1.f, pm1 = 1.f, pm2 = 1.f, pm3 = 1.f;
    {
        const int nb = n0 + oc0;
        if (P.bias) {
            if (nb + 0 < P.N) pb0 = P.bias[nb + 0];
            if (nb + 1 < P.N) pb1 = P.bias[nb + 1];
            if (nb + 2 < P.N) pb2 = P.bias[nb + 2];
            if (nb + 3 < P.N) pb3 = P.bias[nb + 3];
        }
        if (P.mask && m < P.M) {
            const float* mk = P.mask + (size_t)m * P.ldc + nb;
            if (nb + 0 < P.N) pm0 = mk[0];
            if (nb + 1 < P.N) pm1 = mk[1];
            if (nb + 2 < P.N) pm2 = mk[2];
            if (nb + 3 < P.N) pm3 = mk[3];
        }
    }

    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.f;
    float asum = 0.f;

    if (ALL_VEC) mainloop_co(P, R, m0, n0, lane, wv, stage, acc, asum);
    else switch (P.a_mode * 3 + P.b_mode) {   // workgroup-uniform
        case 0: mainloop<0, 0>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        case 1: mainloop<0, 1>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        case 2: mainloop<0, 2>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        case 3: mainloop<1, 0>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        case 4: mainloop<1, 1>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        case 5: mainloop<1, 2>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        case 6: mainloop<2, 0>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        case 7: mainloop<2, 1>(P, R, m0, n0, i, kh, wv, acc, asum); break;
        default: mainloop<2, 2>(P, R, m0, n0, i, kh, wv, acc, asum); break;
    }

    if (stop_taken(stopv)) return;
    // ---- combine the 4 K-partials through LDS (fixed order => deterministic) -------------
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int r = (s & 3) + 8 * (s >> 2) + 4 * kh;
        red[wv][r * 33 + i] = acc[s];
    }
    dbr[wv * 2 + kh][i] = asum;
    __syncthreads();

    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = n0 + oc0 + e;
        float v = ((red[0][orow * 33 + oc0 + e] + red[1][orow * 33 + oc0 + e]) +
                   red[2][orow * 33 + oc0 + e]) + red[3][orow * 33 + oc0 + e];
        const float pb = e == 0 ? pb0 : e == 1 ? pb1 : e == 2 ? pb2 : pb3;
        const float pm = e == 0 ? pm0 : e == 1 ? pm1 : e == 2 ? pm2 : pm3;
        if (m < P.M && n < P.N) {
            if (P.bias) v += pb;
            v = act_f(v, P.act);
            if (P.mask) v = (pm > 0.f) ? v : 0.f;
            P.C[(size_t)sp * P.c_split + (size_t)m * P.ldc + n] = v;
            if (P.CT) P.CT[(size_t)n * P.ldct + m] = v;
            ss += v * v;
        }
    }
    if (P.dbias && tn == 0 && tid < 32) {
        float d = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) d += dbr[w][tid];
        if (m0 + tid < P.M) {
            P.dbias[(size_t)sp * P.M + m0 + tid] = d;
            ss += d * d;
        }
    }
    if (P.sumsq) {
        const float t = smx_block_sum(ss, sred);
        if (tid == 0) P.sumsq[tile] = t;
    }
}

template <bool ALL_VEC>
__global__ __launch_bounds__(256, 2) void gemm32_kernel(GemmBatch G) {
    gemm32_body<ALL_VEC>(G);
}

// Many-row variant (M >= 2048: the stem passes over B*T or frames*pixels rows).  A workgroup owns a
// 128 x 32 output tile and its four waves split the ROWS: each wave walks all of K for its own
// 32 x 32 tile and stores it straight from the accumulators -- no cross-wave reduction, and the
// per-workgroup fixed cost is spread over four times the work.  No bias-gradient / sum-of-squares
// outputs (the weight gradients take the K-split kernel).
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(GemmBatch G) {
    const TileBases TB = load_tile_bases(G);
    const GemmProb P = select_problem(G, TB, (int)blockIdx.x);
    const int stopv = stop_load(P);
    const int tile = blockIdx.x - P.tile_base;
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int m0 = (tm * 4 + wv) * 32, n0 = tn * 32;
    if (m0 >= P.M) return;
    KRange R;
    R.A = P.A; R.B = P.B; R.a_bytes = P.a_bytes; R.b_bytes = P.b_bytes; R.K = P.K;
    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.f;
    float asum = 0.f;
    switch (P.a_mode * 3 + P.b_mode) {
        case 0: mainloop<0, 0, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        case 1: mainloop<0, 1, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        case 2: mainloop<0, 2, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        case 3: mainloop<1, 0, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        case 4: mainloop<1, 1, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        case 5: mainloop<1, 2, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        case 6: mainloop<2, 0, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        case 7: mainloop<2, 1, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
        default: mainloop<2, 2, 1>(P, R, m0, n0, i, kh, 0, acc, asum); break;
    }
    if (stop_taken(stopv)) return;
    // C fragment: lane holds column n0 + i, reg s holds row (s&3) + 8(s>>2) + 4kh -> each store
    // instruction writes two rows of 32 consecutive floats
    const int n = n0 + i;
    if (n < P.N) {
        const float bias = P.bias ? P.bias[n] : 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int m = m0 + (s & 3) + 8 * (s >> 2) + 4 * kh;
            if (m < P.M) {
                float v = act_f(acc[s] + bias, P.act);
                if (P.mask) v = (P.mask[(size_t)m * P.ldc + n] > 0.f) ? v : 0.f;
                P.C[(size_t)m * P.ldc + n] = v;
                if (P.CT) P.CT[(size_t)n * P.ldct + m] = v;
            }
        }
    }
}

// Throughput variant for the stems' big layers (M >= 2048 rows AND N >= 64: the Linear behind the convolutions,
// 7168 x 2592 x 256, the LSTM's input projection and the MLP on top of it over B*T rows).  gemm_rows_kernel reads both
// operands of every 32 x 32 wave tile straight from L2 in fragment order -- 64 operand words per 1024 MACs, 32 cache
// lines per load instruction -- and sits at 30 - 50 TFLOP/s there.  Here a workgroup owns a (64 TM) x (64 TN) tile, its
// 2 x 2 wavefronts a (32 TM) x (32 TN) quarter each; a 32-wide K block of both operands is fetched as whole 128-byte
// segments (coalesced), staged in LDS (double-buffered: the global loads of block k + 1 are in flight under the MFMAs
// of block k, one barrier per block) and read back as fragments.  The fragment mapping and the K order are those of
// gemm_rows_kernel (pairs (k, k + 4) inside 8-wide groups, blocks ascending), so every output is the SAME sum in the
// same order: the two kernels are bit-identical and the host may pick either by shape.
// Operand storage: MODE 0 (K-contiguous, 16-byte aligned) -> LDS rows of 36 floats, fragments by ds_read_b128;
// MODE 2 (K-strided: X(r, k) = X[k ld + r], ld % 4 == 0, aligned) -> LDS [k][rows + 4], fragments by four ds_read_b32.
template <int MODE, int R>            // R rows of the operand per workgroup tile
struct TileStage {
    static constexpr int NV = R * 32 / 4 / 256;                        // 16-byte words per thread and K block
    static constexpr int LDS_FLOATS = MODE == 0 ? R * 36 : 32 * (R + 4);
    unsigned off[NV];                                                   // byte offset of the word at K block 0, or OOB
    int kk[NV];                                                         // its k inside the block
    int wofs[NV];                                                       // where it goes in the LDS tile (floats)
    unsigned kstep;                                                     // bytes per k
    __device__ __forceinline__ void init(int ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (MODE == 0) {
                const int r = (tid >> 3) + 32 * j, ks = 4 * (tid & 7);
                off[j] = row0 + r < nrows ? ((unsigned)(row0 + r) * (unsigned)ld + (unsigned)ks) * 4u : OOB;
                kk[j] = ks;
                wofs[j] = r * 36 + ks;
            } else {
                constexpr int TPR = R / 4;                              // threads per k row
                const int rs = 4 * (tid % TPR), k = tid / TPR + (256 / TPR) * j;
                off[j] = row0 + rs < nrows ? ((unsigned)k * (unsigned)ld + (unsigned)(row0 + rs)) * 4u : OOB;
                kk[j] = k;
                wofs[j] = k * (R + 4) + rs;
            }
        }
        kstep = MODE == 0 ? 4u : (unsigned)ld * 4u;
    }
    __device__ __forceinline__ void load(u32x4 (&w)[NV], rsrc_t rs, int kb, int K) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool in = kb + kk[j] < K && off[j] != OOB;
            w[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, in ? off[j] + (unsigned)kb * kstep : OOB, 0, 0);
        }
    }
    __device__ __forceinline__ void store(const u32x4 (&w)[NV], float* tile) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) *(u32x4*)(tile + wofs[j]) = w[j];
    }
    // fragment of the 8-wide k group q for the 32 rows at `r0` of the tile: lane (i, kh) <- X(r0 + i, 8 q + 4 kh + 0..3)
    static __device__ __forceinline__ float4 frag(const float* tile, int r0, int i, int kh, int q) {
        if (MODE == 0) return *(const float4*)(tile + (r0 + i) * 36 + 8 * q + 4 * kh);
        const float* t = tile + (8 * q + 4 * kh) * (R + 4) + r0 + i;
        return make_float4(t[0], t[R + 4], t[2 * (R + 4)], t[3 * (R + 4)]);
    }
};

template <int AM, int BM, int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(GemmBatch G) {
    constexpr int RM = 64 * TM, RN = 64 * TN;
    typedef TileStage<AM, RM> SA;
    typedef TileStage<BM, RN> SB;
    extern __shared__ float lds_tile[];
    constexpr int BUF = SA::LDS_FLOATS + SB::LDS_FLOATS;                 // one buffer: [A tile | B tile]
    const TileBases TB = load_tile_bases(G);
    // XCD-aware order: consecutive workgroup ids go round the 8 XCDs; give each XCD a contiguous run of tiles so that
    // the tiles sharing an operand block meet in ONE L2
    const int grid = TB.grid, per = grid >> 3, rem = grid & 7;
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
    const GemmProb P = select_problem(G, TB, bid);
    const int stopv = stop_load(P);
    int tile = bid - P.tile_base;
    // split-K (weight gradients over 10^4 - 10^5 rows): `splits` workgroups share an output tile, each sums its own K
    // chunk into its own slice of C (and of dbias), as in gemm32_kernel
    int sp = 0;
    KRange R;
    R.A = P.A; R.B = P.B; R.a_bytes = P.a_bytes; R.b_bytes = P.b_bytes; R.K = P.K;
    if (P.splits > 1) {
        sp = tile % P.splits;
        tile /= P.splits;
        const int k0 = sp * P.k_chunk;
        R.K = min(P.k_chunk, P.K - k0);
        R.A = P.A + (P.a_kc ? (size_t)k0 : (size_t)k0 * P.lda);
        R.B = P.B + (P.b_kc ? (size_t)k0 : (size_t)k0 * P.ldb);
        R.a_bytes = 4u * (P.a_kc ? (unsigned)(P.M - 1) * P.lda + R.K : (unsigned)(R.K - 1) * P.lda + P.M);
        R.b_bytes = 4u * (P.b_kc ? (unsigned)(P.N - 1) * P.ldb + R.K : (unsigned)(R.K - 1) * P.ldb + P.N);
    }
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int m0 = tm * RM, n0 = tn * RN;
    const int wm0 = (wv >> 1) * 32 * TM, wn0 = (wv & 1) * 32 * TN;       // this wave's quarter inside the tile
    const rsrc_t ra = make_rsrc(R.A, R.a_bytes), rb = make_rsrc(R.B, R.b_bytes);
    SA sa;
    SB sb;
    sa.init(P.lda, m0, P.M, tid);
    sb.init(P.ldb, n0, P.N, tid);
    float asum[TM];                                                       // row sums of A (the bias gradient)
#pragma unroll
    for (int a = 0; a < TM; ++a) asum[a] = 0.f;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nkb = (R.K + 31) >> 5;
    // One K block of look-ahead: the global loads of block k + 1 are in flight under the MFMAs of block k.  (A three-block
    // register ring was measured and dropped: 118 instead of 70 VGPRs, 4 instead of 7 wavefronts per SIMD, 3 % slower on
    // the many-workgroup shapes and no faster on the few-workgroup ones -- with one wavefront per SIMD the block time,
    // 2400 cycles against 1090 of MFMA issue, is LDS round trips, the dependent accumulator chain and the barrier.)
    u32x4 wa[SA::NV], wb[SB::NV];
    sa.load(wa, ra, 0, R.K);
    sb.load(wb, rb, 0, R.K);
    sa.store(wa, lds_tile);
    sb.store(wb, lds_tile + SA::LDS_FLOATS);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const float* tA = lds_tile + (kb & 1) * BUF;
        const float* tB = tA + SA::LDS_FLOATS;
        if (kb + 1 < nkb) {                                              // wave-uniform
            sa.load(wa, ra, (kb + 1) * 32, R.K);
            sb.load(wb, rb, (kb + 1) * 32, R.K);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = SA::frag(tA, wm0 + 32 * a, i, kh, q);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = SB::frag(tB, wn0 + 32 * b, i, kh, q);
#pragma unroll
            for (int a = 0; a < TM; ++a) asum[a] += (fa[a].x + fa[a].y) + (fa[a].z + fa[a].w);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = MFMA32(fa[a].x, fb[b].x, acc[a][b]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = MFMA32(fa[a].y, fb[b].y, acc[a][b]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = MFMA32(fa[a].z, fb[b].z, acc[a][b]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = MFMA32(fa[a].w, fb[b].w, acc[a][b]);
        }
        if (kb + 1 < nkb) {
            float* nA = lds_tile + ((kb + 1) & 1) * BUF;
            sa.store(wa, nA);
            sb.store(wb, nA + SA::LDS_FLOATS);
        }
        __syncthreads();
    }
    if (stop_taken(stopv)) return;
    if (P.dbias && tn == 0 && wn0 == 0) {                                // wave-uniform
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const float d = asum[a] + __shfl_xor(asum[a], 32, 64);        // the two k halves of a row
            const int m = m0 + wm0 + 32 * a + i;
            if (kh == 0 && m < P.M) P.dbias[(size_t)sp * P.M + m] = d;
        }
    }
    float* const Cs = P.C + (size_t)sp * P.c_split;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn0 + 32 * b + i;
        if (n >= P.N) continue;
        const float bias = P.bias ? P.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < P.M) {
                    float v = act_f(acc[a][b][r] + bias, P.act);
                    if (P.mask) v = (P.mask[(size_t)m * P.ldc + n] > 0.f) ? v : 0.f;
                    Cs[(size_t)m * P.ldc + n] = v;
                    if (P.CT) P.CT[(size_t)n * P.ldct + m] = v;
                }
            }
    }
}

inline unsigned long long operand_bytes(int ld, int kc, int nrows, int K) {
    return kc ? 4ull * ((unsigned long long)(nrows - 1) * ld + K)
              : 4ull * ((unsigned long long)(K - 1) * ld + nrows);
}

inline void fill_prob(GemmProb& P, const float* A, int lda, int a_kc, const float* B, int ldb,
                      int b_kc, const float* bias, const float* mask, float* C, int ldc, int M,
                      int N, int K, int act, float* dbias, float* sumsq, int tile_base,
                      const int* stop = nullptr, float* CT = nullptr, int ldct = 0) {
    P.A = A; P.B = B; P.bias = bias; P.mask = mask; P.C = C; P.dbias = dbias; P.sumsq = sumsq;
    P.CT = CT; P.ldct = ldct;
    P.stop = stop;
    P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.M = M; P.N = N; P.K = K;
    P.a_kc = a_kc; P.b_kc = b_kc; P.act = act;
    P.tiles_m = (M + 31) / 32; P.tiles_n = (N + 31) / 32; P.tile_base = tile_base;
    P.splits = 1; P.k_chunk = K; P.c_split = 0;
    P.a_mode = !a_kc ? 2 : ((lda % 4 == 0 && K % 4 == 0 && ((uintptr_t)A & 15) == 0) ? 0 : 1);
    P.b_mode = !b_kc ? 2 : ((ldb % 4 == 0 && K % 4 == 0 && ((uintptr_t)B & 15) == 0) ? 0 : 1);
    P.a_bytes = (unsigned)operand_bytes(lda, a_kc, M, K);
    P.b_bytes = (unsigned)operand_bytes(ldb, b_kc, N, K);
}

// every operand must fit a 31-bit byte offset (buffer descriptor addressing)
inline bool prob_ok(const GemmProb& P) {
    const int Kc = P.splits > 1 ? P.k_chunk : P.K;       // what one workgroup addresses
    return operand_bytes(P.lda, P.a_kc, P.M, Kc) < (1ull << 31) &&
           operand_bytes(P.ldb, P.b_kc, P.N, Kc) < (1ull << 31);
}

// gemm_tile_kernel: eligible when every problem of the batch has the same operand storage out of {K-contiguous aligned,
// K-strided with ld % 4 == 0 and an aligned base}, N >= 64 and K >= 32.
template <int AM, int BM, int TM, int TN>
inline int launch_tile_variant(GemmBatch& G, hipStream_t st) {
    int base = 0;
    for (int k = 0; k < G.n; ++k) {
        G.p[k].tiles_m = (G.p[k].M + 64 * TM - 1) / (64 * TM);
        G.p[k].tiles_n = (G.p[k].N + 64 * TN - 1) / (64 * TN);
        G.p[k].tile_base = base;
        base += G.p[k].tiles_m * G.p[k].tiles_n * G.p[k].splits;
    }
    const size_t lds = 2 * sizeof(float) * (size_t)(TileStage<AM, 64 * TM>::LDS_FLOATS + TileStage<BM, 64 * TN>::LDS_FLOATS);
    auto kern = gemm_tile_kernel<AM, BM, TM, TN>;
    static bool attr_done = false;                        // per instantiation
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(base), dim3(256), lds, st, G);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}

// Tile shape: 64 x 64 (TM = TN = 1, 32 x 32 per wavefront).  Measured on the stems' shapes (scripts/bench_gemm_tile.py):
// 7168 x 256 x 2592: 85 TFLOP/s against 81 (128 x 64) and 48 (128 x 128); 7168 x 2592 x 256: 75 / 54 / 52; 131072 x 300 x
// 100: 57 / 48 / 42 -- with one K block of look-ahead the bigger tiles' 2 - 3 workgroups per CU cannot cover a workgroup's
// load -> LDS -> barrier bubble, the small tile's 6 - 7 can.  (gemm_rows_kernel on the same three: 59 / 47 / 35.)
template <int AM, int BM>
inline int launch_tile_modes(GemmBatch& G, hipStream_t st) {
    return launch_tile_variant<AM, BM, 1, 1>(G, st);
}

inline bool tile_mode_of(const float* X, int ld, int kc, int mode, int& out) {
    if (kc) { out = 0; return mode == 0; }
    out = 2;
    return ld % 4 == 0 && ((uintptr_t)X & 15) == 0;
}

// -> true when the batch was taken (rc = the launch's result)
inline bool launch_tiles(GemmBatch& G, hipStream_t st, int& rc) {
    static const bool off = getenv("SMX_GEMM_ROWS_ONLY") != nullptr;     // A/B switch for measurements
    if (off) return false;
    int am = -1, bm = -1;
    for (int k = 0; k < G.n; ++k) {
        const GemmProb& P = G.p[k];
        int a, b;
        if (!tile_mode_of(P.A, P.lda, P.a_kc, P.a_mode, a) || !tile_mode_of(P.B, P.ldb, P.b_kc, P.b_mode, b)) return false;
        if (P.N < 64 || (P.splits > 1 ? P.k_chunk : P.K) < 32 || P.sumsq) return false;
        if (k && (a != am || b != bm)) return false;
        am = a; bm = b;
    }
    if (am == 0 && bm == 0) rc = launch_tile_modes<0, 0>(G, st);
    else if (am == 0 && bm == 2) rc = launch_tile_modes<0, 2>(G, st);
    else if (am == 2 && bm == 0) rc = launch_tile_modes<2, 0>(G, st);
    else rc = launch_tile_modes<2, 2>(G, st);
    return true;
}

// would launch_batch() hand this weight-gradient batch to the 64 x 64 tile kernel (below: `wide && wgs >= 512` and the
// tile kernel's operand rules)?  Callers that MERGE batches ask first: a merge must not move a problem to another kernel
// (another summation order).
inline bool wgrad_batch_takes_tiles(const GemmBatch& G) {
    static const bool off = getenv("SMX_GEMM_ROWS_ONLY") != nullptr;
    if (off) return false;
    bool wide = true;
    long wgs = 0;
    int am = -1, bm = -1;
    for (int k = 0; k < G.n; ++k) {
        const GemmProb& P = G.p[k];
        wide = wide && P.K >= 2048 && P.M >= 48 && !P.sumsq;
        wgs += (long)((P.M + 63) / 64) * ((P.N + 63) / 64) * P.splits;
        int a, b;
        if (!tile_mode_of(P.A, P.lda, P.a_kc, P.a_mode, a) || !tile_mode_of(P.B, P.ldb, P.b_kc, P.b_mode, b)) return false;
        if (P.N < 64 || (P.splits > 1 ? P.k_chunk : P.K) < 32 || P.sumsq) return false;
        if (k && (a != am || b != bm)) return false;
        am = a; bm = b;
    }
    return wide && wgs >= 512;
}
// problems of `src` appended to `dst` (tile bases renumbered)
inline void batch_append(GemmBatch& dst, const GemmBatch& src) {
    int base = 0;
    if (dst.n) {
        const GemmProb& L = dst.p[dst.n - 1];
        base = L.tile_base + L.tiles_m * L.tiles_n * L.splits;
    }
    for (int k = 0; k < src.n; ++k) {
        GemmProb& P = dst.p[dst.n++];
        P = src.p[k];
        P.tile_base = base;
        base += P.tiles_m * P.tiles_n * P.splits;
    }
}

inline int launch_batch(GemmBatch& G, hipStream_t st) {
    for (int k = 0; k < G.n; ++k)
        if (!prob_ok(G.p[k])) return SMX_E_SHAPE;
    bool rows_variant = true;
    int rc_tiles = SMX_OK;
    for (int k = 0; k < G.n; ++k)
        rows_variant = rows_variant && G.p[k].M >= 2048 && !G.p[k].dbias && !G.p[k].sumsq &&
                       G.p[k].splits == 1;
    if (rows_variant && launch_tiles(G, st, rc_tiles)) return rc_tiles;
    // weight gradients over many rows (M x N = the weight's shape, K = rows >= 2048, split or not): 64 x 64 tiles read
    // each operand word for 64 outputs instead of 32 -- at 10^5 rows the 32 x 32 kernel is bound by L2 -> CU traffic
    {
        bool wide = true;
        long wgs = 0;
        for (int k = 0; k < G.n; ++k) {
            wide = wide && G.p[k].K >= 2048 && G.p[k].M >= 48 && !G.p[k].sumsq;
            wgs += (long)((G.p[k].M + 63) / 64) * ((G.p[k].N + 63) / 64) * G.p[k].splits;
        }
        // ... when there are enough of them: a workgroup alone on its CU runs a K block in 2400 cycles (1090 of MFMA
        // issue); below ~2 workgroups per CU the 32 x 32 kernel's four-way K split inside the workgroup is faster
        // (measured: 7936 rows, dW_hh 98 workgroups 36 us against 20 us; 127 k rows, 896: 0.19 against 0.38 ms)
        if (wide && wgs >= 512 && launch_tiles(G, st, rc_tiles)) return rc_tiles;
    }
    if (rows_variant) {
        int base = 0;
        for (int k = 0; k < G.n; ++k) {
            G.p[k].tiles_m = (G.p[k].M + 127) / 128;        // 128-row workgroup tiles
            G.p[k].tile_base = base;
            base += G.p[k].tiles_m * G.p[k].tiles_n;
        }
        hipLaunchKernelGGL(gemm_rows_kernel, dim3(base), dim3(256), 0, st, G);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? SMX_OK : (int)e;
    }
    const GemmProb& L = G.p[G.n - 1];
    const int blocks = L.tile_base + L.tiles_m * L.tiles_n * L.splits;
    bool all_vec = true;
    for (int k = 0; k < G.n; ++k) all_vec = all_vec && G.p[k].a_mode == 0 && G.p[k].b_mode == 0;
    if (all_vec) hipLaunchKernelGGL(gemm32_kernel<true>, dim3(blocks), dim3(256), 0, st, G);
    else hipLaunchKernelGGL(gemm32_kernel<false>, dim3(blocks), dim3(256), 0, st, G);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}

}  // namespace

extern "C" int smx_linear_f32(const float* A, int32_t lda, int32_t a_kcontig, const float* B,
                              int32_t ldb, int32_t b_kcontig, const float* bias, float* C,
                              int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t act,
                              const float* relu_mask, const int32_t* stop_flag,
                              smx_stream_t stream) {
    SMX_REQUIRE(A && B && C, SMX_E_NULL);
    SMX_REQUIRE(M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N, SMX_E_SHAPE);
    GemmBatch G;
    // An operand is addressed through a buffer descriptor (31-bit byte offsets).  A row-major A
    // past 2 GiB -- the patch matrix of a convolution over thousands of frames -- is cut into
    // row blocks, one problem of the same launch each (rows are independent).
    const unsigned long long abytes = operand_bytes(lda, a_kcontig, M, K);
    int parts = 1;
    if (a_kcontig && abytes >= (1ull << 31)) parts = (int)(abytes / ((1ull << 31) - (1ull << 20))) + 1;
    SMX_REQUIRE(parts <= MAX_PROBS, SMX_E_UNSUPPORTED);
    const int rows_per = parts == 1 ? M : ((((M + parts - 1) / parts) + 127) & ~127);
    G.n = 0;
    int base = 0;
    for (int m_off = 0; m_off < M; m_off += rows_per) {
        const int Mp = M - m_off < rows_per ? M - m_off : rows_per;
        GemmProb& P = G.p[G.n++];
        fill_prob(P, A + (size_t)m_off * lda, lda, a_kcontig, B, ldb, b_kcontig, bias,
                  relu_mask ? relu_mask + (size_t)m_off * ldc : nullptr, C + (size_t)m_off * ldc, ldc, Mp,
                  N, K, act, nullptr, nullptr, base, stop_flag);
        base += P.tiles_m * P.tiles_n;
    }
    return launch_batch(G, smx_s(stream));
}

// Up to 9 INDEPENDENT dense problems in one launch (the latency of a short launch is paid once per dependency level
// instead of once per layer): each job is either a layer  C = act(A . B^T + bias) [* (mask > 0)]  (kind 0, the
// arguments of smx_linear_f32) or a weight gradient  dW = dZ^T . X, db = column sums of dZ  (kind 1, the arguments of
// smx_linear_wgrad_f32: A = dZ, B = X, C = dW, dbias = db, M x N = dW's shape, K = rows).
extern "C" int smx_linear_multi_f32(const smx_linear_job_t* jobs, int32_t njobs, smx_stream_t stream) {
    SMX_REQUIRE(jobs, SMX_E_NULL);
    SMX_REQUIRE(njobs >= 1 && njobs <= MAX_PROBS, SMX_E_SHAPE);
    GemmBatch G;
    G.n = njobs;
    int base = 0;
    for (int k = 0; k < njobs; ++k) {
        const smx_linear_job_t& j = jobs[k];
        SMX_REQUIRE(j.A && j.B && j.C, SMX_E_NULL);
        SMX_REQUIRE(j.M > 0 && j.N > 0 && j.K > 0 && j.lda > 0 && j.ldb > 0 && j.ldc >= j.N, SMX_E_SHAPE);
        if (j.kind == 1) {
            SMX_REQUIRE(j.lda >= j.M && j.ldb >= j.N, SMX_E_SHAPE);
            fill_prob(G.p[k], j.A, j.lda, 0, j.B, j.ldb, 0, nullptr, nullptr, j.C, j.ldc, j.M, j.N, j.K, SMX_ACT_NONE,
                      j.dbias, nullptr, base, j.stop_flag);
        } else {
            SMX_REQUIRE(j.kind == 0, SMX_E_UNSUPPORTED);
            fill_prob(G.p[k], j.A, j.lda, j.a_kcontig, j.B, j.ldb, j.b_kcontig, j.bias, j.relu_mask, j.C, j.ldc, j.M, j.N,
                      j.K, j.act, nullptr, nullptr, base, j.stop_flag);
        }
        base += G.p[k].tiles_m * G.p[k].tiles_n;
    }
    return launch_batch(G, smx_s(stream));
}

// dW[M,N] = dZ^T . X (dZ [rows, >=M] stride ldz, X [rows, >=N] stride ldx), db[M] = column sums of dZ
extern "C" int smx_linear_wgrad_f32(const float* dZ, int32_t ldz, const float* X, int32_t ldx,
                                    float* dW, int32_t ldw, float* db, int32_t M, int32_t N,
                                    int32_t rows, smx_stream_t stream) {
    SMX_REQUIRE(dZ && X && dW, SMX_E_NULL);
    SMX_REQUIRE(M > 0 && N > 0 && rows > 0 && ldz >= M && ldx >= N && ldw >= N, SMX_E_SHAPE);
    GemmBatch G;
    G.n = 1;
    fill_prob(G.p[0], dZ, ldz, 0, X, ldx, 0, nullptr, nullptr, dW, ldw, M, N, rows, SMX_ACT_NONE, db,
              nullptr, 0);
    return launch_batch(G, smx_s(stream));
}

// ---------------------------------------------------------------------------
// split-K weight gradient: with rows >> 10^3 a 32x32 tile of dW would be ONE workgroup walking
// all rows (LSTM / CNN stems: rows = B*T, B*E*pixels ~ 10^5).  The rows are cut into `splits`
// chunks, each (tile, chunk) is a workgroup writing a partial tile into the caller's workspace,
// and a second launch adds the chunks in a fixed order (deterministic).
// ---------------------------------------------------------------------------
namespace {

// Partial sums of SEVERAL split-K problems reduced in one launch: segment g is `count` consecutive elements whose
// partial s sits at src + s * stride; 16 elements x 16 slices of the split index per workgroup, slices combined through
// LDS in slice order (deterministic).
struct RedSeg {
    const float* src;
    float* dst;
    int base, stride;
};
struct RedSegs {
    RedSeg g[6];
    int n, total;
};
__global__ __launch_bounds__(256) void segmented_reduce_kernel(RedSegs L, int splits, const int* __restrict__ stop) {
    if (stop && *stop) return;
    __shared__ float red[16][16];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const bool valid = e < L.total;
    int gi = 0;
#pragma unroll
    for (int k = 1; k < 6; ++k) gi += (k < L.n && e >= L.g[k].base) ? 1 : 0;
    const RedSeg G = L.g[gi];
    const float* src = G.src + (e - G.base);
    const int per = (splits + 15) >> 4;
    const int s0 = sl * per, s1 = s0 + per < splits ? s0 + per : splits;
    float v0 = 0.f, v1 = 0.f;
    if (valid) {
        int sidx = s0;
        for (; sidx + 1 < s1; sidx += 2) {
            v0 += src[(size_t)sidx * G.stride];
            v1 += src[(size_t)(sidx + 1) * G.stride];
        }
        if (sidx < s1) v0 += src[(size_t)sidx * G.stride];
    }
    red[sl][el] = v0 + v1;
    __syncthreads();
    if (sl == 0 && valid) {
        float v = red[0][el];
#pragma unroll
        for (int j = 1; j < 16; ++j) v += red[j][el];
        G.dst[e - G.base] = v;
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part,
                                                            const float* __restrict__ dbpart,
                                                            int splits, int M, int N,
                                                            float* __restrict__ dW, int ldw,
                                                            float* __restrict__ db) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total + M; i += (long)gridDim.x * 256) {
        if (i < total) {
            float v = 0.f;
            for (int s = 0; s < splits; ++s) v += part[(size_t)s * total + i];
            const long m = i / N;
            dW[m * ldw + (i - m * N)] = v;
        } else if (db) {
            const long m = i - total;
            float v = 0.f;
            for (int s = 0; s < splits; ++s) v += dbpart[(size_t)s * M + m];
            db[m] = v;
        }
    }
}

inline int pick_splits(int M, int N, int rows) {
    int gw;
    if (rows >= SMX_WGRAD_ROWS_MIN && M % 4 == 0 && N % 4 == 0 && smx_wgrad_rows_groups(M, N, &gw) > 0) {
        const int s = rows / 128;           // smx_wgrad.hip: one workgroup per CU walks one chunk of >= 128 rows
        return s > 256 ? 256 : s;
    }
    const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
    int s = rows / 1024;                    // >= 8 super-blocks per wave and chunk
    const int cap = (2048 + tiles - 1) / tiles;     // ~8 workgroups per CU at most
    if (s > cap) s = cap;
    if (s > 256) s = 256;
    return s < 2 ? 1 : s;
}

// the second launch of a split-K weight gradient: the chunks' partial tiles added in chunk order
inline int splitk_reduce_launch(float* part, float* dbpart, int splits, int M, int N, float* dW, int ldw, float* db,
                                hipStream_t st) {
    const long total = (long)M * N + M;
    if (ldw == N && splits > 32) {
        // many chunks: 16 slices of the split index per element in parallel, combined in slice order (one thread per
        // element walking 248 partials measured 67 us for a 400 x 100 gradient)
        RedSegs L;
        L.n = db ? 2 : 1;
        L.g[0] = RedSeg{part, dW, 0, M * N};
        L.g[1] = RedSeg{dbpart, db, M * N, M};
        L.total = db ? (int)total : M * N;
        hipLaunchKernelGGL(segmented_reduce_kernel, dim3((unsigned)((L.total + 15) / 16)), dim3(256), 0, st, L, splits,
                           (const int*)nullptr);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? SMX_OK : (int)e;
    }
    long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, dbpart, splits, M, N, dW, ldw, db);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}

}  // namespace

extern "C" int64_t smx_linear_wgrad_ws_floats(int32_t M, int32_t N, int32_t rows) {
    const int s = pick_splits(M, N, rows);
    return s > 1 ? (int64_t)s * ((int64_t)M * N + M) : 0;
}

extern "C" int smx_linear_wgrad_splitk_f32(const float* dZ, int32_t ldz, const float* X,
                                           int32_t ldx, float* dW, int32_t ldw, float* db,
                                           int32_t M, int32_t N, int32_t rows, float* ws,
                                           int64_t ws_floats, smx_stream_t stream) {
    SMX_REQUIRE(dZ && X && dW, SMX_E_NULL);
    SMX_REQUIRE(M > 0 && N > 0 && rows > 0 && ldz >= M && ldx >= N && ldw >= N, SMX_E_SHAPE);
    const int S = pick_splits(M, N, rows);
    if (S <= 1 || !ws || ws_floats < (int64_t)S * ((int64_t)M * N + M))
        return smx_linear_wgrad_f32(dZ, ldz, X, ldx, dW, ldw, db, M, N, rows, stream);
    GemmBatch G;
    G.n = 1;
    float* part = ws;
    float* dbpart = ws + (size_t)S * M * N;
    fill_prob(G.p[0], dZ, ldz, 0, X, ldx, 0, nullptr, nullptr, part, N, M, N, rows, SMX_ACT_NONE,
              dbpart, nullptr, 0);
    G.p[0].splits = S;
    G.p[0].k_chunk = ((rows + S - 1) / S + 31) & ~31;
    G.p[0].c_split = (long)M * N;
    // every chunk must be non-empty
    while ((long)(G.p[0].splits - 1) * G.p[0].k_chunk >= rows) --G.p[0].splits;
    static const bool tiled_only = getenv("SMX_WGRAD_TILED") != nullptr;
    if (!tiled_only && smx_wgrad_rows_eligible(dZ, ldz, X, ldx, M, N, rows)) {
        // the whole dW in one workgroup's registers, every operand row read once (smx_wgrad.hip)
        // (a dW wider than one workgroup's registers -- the LSTM's dW_ih at 376 inputs: 400 x 376 -- goes as column groups,
        // each a problem of the same launch that re-reads dZ)
        WgradBatch Gd;
        int gw = N;
        Gd.n = smx_wgrad_rows_groups(M, N, &gw);
        Gd.stop = nullptr;
        for (int gi = 0; gi < Gd.n; ++gi) {
            WgradProb& P = Gd.p[gi];
            const int c0 = gi * gw;
            P.A = dZ; P.B = X + c0; P.Cpart = part + c0; P.bpart = gi == 0 ? dbpart : nullptr;
            P.M = M; P.N = (N - c0 < gw) ? N - c0 : gw; P.lda = ldz; P.ldb = ldx; P.rows = rows;
            P.ldc = N; P.c_split = (long)M * N; P.b_col0 = c0;
            P.splits = G.p[0].splits; P.k_chunk = G.p[0].k_chunk;
        }
        const int rc = smx_wgrad_rows_launch(Gd, smx_s(stream));
        if (rc) return rc;
    } else {
        const int rc = launch_batch(G, smx_s(stream));
        if (rc) return rc;
    }
    return splitk_reduce_launch(part, dbpart, G.p[0].splits, M, N, dW, ldw, db, smx_s(stream));
}

extern "C" int smx_linear_wgrad_splitk_pair_f32(const float* dZ, int32_t ldz, int32_t M, int32_t rows, const float* X1,
                                                int32_t ldx1, float* dW1, float* db1, int32_t N1, const float* X2,
                                                int32_t ldx2, float* dW2, float* db2, int32_t N2, float* ws,
                                                int64_t ws_floats, smx_stream_t stream) {
    SMX_REQUIRE(dZ && X1 && dW1 && X2 && dW2, SMX_E_NULL);
    SMX_REQUIRE(M > 0 && N1 > 0 && N2 > 0 && rows > 0 && ldz >= M && ldx1 >= N1 && ldx2 >= N2, SMX_E_SHAPE);
    const int S1 = pick_splits(M, N1, rows), S2 = pick_splits(M, N2, rows);
    const int64_t need1 = S1 > 1 ? (int64_t)S1 * ((int64_t)M * N1 + M) : 0, need2 = S2 > 1 ? (int64_t)S2 * ((int64_t)M * N2 + M) : 0;
    static const bool tiled_only = getenv("SMX_WGRAD_TILED") != nullptr;
    static const bool no_pair = getenv("SMX_WGRAD_NO_PAIR") != nullptr;       // A/B switch
    bool pair = !no_pair && S1 > 1 && S2 > 1 && ws && ws_floats >= need1 + need2;
    pair = pair && (tiled_only || (!smx_wgrad_rows_eligible(dZ, ldz, X1, ldx1, M, N1, rows) &&
                                   !smx_wgrad_rows_eligible(dZ, ldz, X2, ldx2, M, N2, rows)));
    GemmBatch G, G1, G2;
    float* part[2] = {ws, ws + need1};
    if (pair) {
        const float* X[2] = {X1, X2};
        const int ldx[2] = {ldx1, ldx2}, N[2] = {N1, N2}, S[2] = {S1, S2};
        G.n = 0;
        int base = 0;
        for (int k = 0; k < 2; ++k) {
            GemmProb& P = G.p[G.n++];
            fill_prob(P, dZ, ldz, 0, X[k], ldx[k], 0, nullptr, nullptr, part[k], N[k], M, N[k], rows, SMX_ACT_NONE,
                      part[k] + (size_t)S[k] * M * N[k], nullptr, base);
            P.splits = S[k];
            P.k_chunk = ((rows + S[k] - 1) / S[k] + 31) & ~31;
            P.c_split = (long)M * N[k];
            while ((long)(P.splits - 1) * P.k_chunk >= rows) --P.splits;      // every chunk non-empty
            base += P.tiles_m * P.tiles_n * P.splits;
        }
        G1.n = G2.n = 1;
        G1.p[0] = G.p[0];
        G2.p[0] = G.p[1]; G2.p[0].tile_base = 0;
        // each problem must stay on the kernel it has alone (the 32 x 32 one)
        pair = !wgrad_batch_takes_tiles(G1) && !wgrad_batch_takes_tiles(G2) && !wgrad_batch_takes_tiles(G);
    }
    if (!pair) {
        const int rc = smx_linear_wgrad_splitk_f32(dZ, ldz, X1, ldx1, dW1, N1, db1, M, N1, rows, ws, ws_floats, stream);
        if (rc) return rc;
        return smx_linear_wgrad_splitk_f32(dZ, ldz, X2, ldx2, dW2, N2, db2, M, N2, rows, ws, ws_floats, stream);
    }
    int rc = launch_batch(G, smx_s(stream));
    if (rc) return rc;
    rc = splitk_reduce_launch(part[0], part[0] + (size_t)S1 * M * N1, G.p[0].splits, M, N1, dW1, N1, db1, smx_s(stream));
    if (rc) return rc;
    return splitk_reduce_launch(part[1], part[1] + (size_t)S2 * M * N2, G.p[1].splits, M, N2, dW2, N2, db2, smx_s(stream));
}

// ---------------------------------------------------------------------------
// multi-job MLP forward / backward: the layer-l GEMMs of up to MAX_JOBS independent networks
// (PPO: actor + critic, which the reference updates in two separate loops, ppo.py:541-562)
// share ONE launch per layer -- half the launches, twice the workgroups per launch.
// ---------------------------------------------------------------------------
constexpr int MAX_JOBS = 3;       // backward: 3 GEMM problems per job in one launch (the third job:
                                  // the actor's KL right-hand side of a data-parallel epoch)
constexpr int MAX_FWD_JOBS = 4;   // forward: 1 problem per job and layer (e.g. actor, critic,
                                  // reference actor and the critic's obs_next rows of a learn)

extern "C" int smx_mlp3_forward_multi_f32(const smx_mlp3_job_t* jobs, int32_t njobs,
                                          smx_stream_t stream) {
    SMX_REQUIRE(jobs, SMX_E_NULL);
    SMX_REQUIRE(njobs >= 1 && njobs <= MAX_FWD_JOBS, SMX_E_SHAPE);
    for (int j = 0; j < njobs; ++j) {
        const smx_mlp3_job_t& J = jobs[j];
        SMX_REQUIRE(J.net && J.x && J.h1 && J.h2 && J.out, SMX_E_NULL);
        SMX_REQUIRE(J.out_ld == 0 || J.out_ld >= J.net->OUT, SMX_E_SHAPE);
        SMX_REQUIRE(J.rows > 0 && J.rows < (1 << 30), SMX_E_SHAPE);
    }
    for (int layer = 0; layer < 3; ++layer) {
        GemmBatch G;
        G.n = njobs;
        int base = 0;
        for (int j = 0; j < njobs; ++j) {
            const smx_mlp3_job_t& J = jobs[j];
            const smx_mlp3_t* n = J.net;
            const int R = (int)J.rows;
            if (layer == 0)
                fill_prob(G.p[j], J.x, n->D, 1, n->W1, n->D, 1, n->b1, nullptr, J.h1, n->H1, R, n->H1,
                          n->D, SMX_ACT_RELU, nullptr, nullptr, base, J.stop_flag, J.h1T, (int)(J.ldT ? J.ldT : R));
            else if (layer == 1)
                fill_prob(G.p[j], J.h1, n->H1, 1, n->W2, n->H1, 1, n->b2, nullptr, J.h2, n->H2, R,
                          n->H2, n->H1, SMX_ACT_RELU, nullptr, nullptr, base, J.stop_flag, J.h2T, (int)(J.ldT ? J.ldT : R));
            else
                fill_prob(G.p[j], J.h2, n->H2, 1, n->W3, n->H2, 1, n->b3, nullptr, J.out,
                          J.out_ld ? J.out_ld : n->OUT, R, n->OUT, n->H2, J.out_act, nullptr, nullptr,
                          base, J.stop_flag);
            base += G.p[j].tiles_m * G.p[j].tiles_n;
        }
        const int rc = launch_batch(G, smx_s(stream));
        if (rc) return rc;
    }
    return SMX_OK;
}

extern "C" int smx_mlp3_forward_f32(const smx_mlp3_t* net, const float* x, int64_t rows, float* h1,
                                    float* h2, float* out, int32_t out_act,
                                    const int32_t* stop_flag, smx_stream_t stream) {
    smx_mlp3_job_t J;
    memset(&J, 0, sizeof(J));
    J.net = net; J.x = x; J.rows = rows; J.h1 = h1; J.h2 = h2; J.out = out; J.out_act = out_act;
    J.stop_flag = stop_flag;
    SMX_REQUIRE(net, SMX_E_NULL);
    return smx_mlp3_forward_multi_f32(&J, 1, stream);
}

extern "C" int32_t smx_mlp3_backward_partials(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    auto t = [](int a) { return (a + 31) / 32; };
    return t(H1) * t(D) + t(H2) * t(H1) + t(OUT) * t(H2);
}

// dW_l = dz_l^T . input_l , db_l = column sums of dz_l : 3 problems per job, one launch
static void build_wgrads(GemmBatch& G, const smx_mlp3_job_t* jobs, int32_t njobs) {
    G.n = 3 * njobs;
    int base = 0;
    for (int j = 0; j < njobs; ++j) {
        const smx_mlp3_job_t& J = jobs[j];
        const smx_mlp3_t* n = J.net;
        const int R = (int)J.rows, D = n->D, H1 = n->H1, H2 = n->H2, O = n->OUT;
        float* gW1 = J.grads;
        float* gb1 = gW1 + (size_t)H1 * D;
        float* gW2 = gb1 + H1;
        float* gb2 = gW2 + (size_t)H2 * H1;
        float* gW3 = gb2 + H2;
        float* gb3 = gW3 + (size_t)O * H2;
        float* sq = J.sumsq_partials;
        const int job_base = base;
        // with the transposed copies ([features, rows], written by the producing GEMMs' epilogues)
        // both operands of dW = dz^T . input are K-contiguous: 16-byte fragment loads, 4 MFMA steps
        // per load, instead of one 4-byte load per operand per step
        const bool kc = J.xT && J.h1T && J.h2T && J.dz1T && J.dz2T && J.dz3T;
        const int LT = (int)(J.ldT ? J.ldT : R);
        const float *a1 = kc ? J.dz1T : J.dz1, *a2 = kc ? J.dz2T : J.dz2, *a3 = kc ? J.dz3T : J.dz3;
        const float *b1 = kc ? J.xT : J.x, *b2 = kc ? J.h1T : J.h1, *b3 = kc ? J.h2T : J.h2;
        fill_prob(G.p[3 * j + 0], a1, kc ? LT : H1, kc, b1, kc ? LT : D, kc, nullptr, nullptr, gW1, D, H1,
                  D, R, SMX_ACT_NONE, gb1, sq, base, J.stop_flag);
        base += G.p[3 * j + 0].tiles_m * G.p[3 * j + 0].tiles_n;
        fill_prob(G.p[3 * j + 1], a2, kc ? LT : H2, kc, b2, kc ? LT : H1, kc, nullptr, nullptr, gW2, H1,
                  H2, H1, R, SMX_ACT_NONE, gb2, sq ? sq + (base - job_base) : nullptr, base, J.stop_flag);
        base += G.p[3 * j + 1].tiles_m * G.p[3 * j + 1].tiles_n;
        fill_prob(G.p[3 * j + 2], a3, kc ? LT : O, kc, b3, kc ? LT : H2, kc, nullptr, nullptr, gW3, H2, O,
                  H2, R, SMX_ACT_NONE, gb3, sq ? sq + (base - job_base) : nullptr, base, J.stop_flag);
        base += G.p[3 * j + 2].tiles_m * G.p[3 * j + 2].tiles_n;
    }
}

static int launch_wgrads(const smx_mlp3_job_t* jobs, int32_t njobs, smx_stream_t stream) {
    GemmBatch G;
    build_wgrads(G, jobs, njobs);
    return launch_batch(G, smx_s(stream));
}

extern "C" int smx_mlp3_backward_multi_f32(const smx_mlp3_job_t* jobs, int32_t njobs,
                                           smx_stream_t stream) {
    SMX_REQUIRE(jobs, SMX_E_NULL);
    SMX_REQUIRE(njobs >= 1 && njobs <= MAX_JOBS, SMX_E_SHAPE);
    for (int j = 0; j < njobs; ++j) {
        const smx_mlp3_job_t& J = jobs[j];
        SMX_REQUIRE(J.net && J.x && J.h1 && J.h2 && J.dz3 && J.dz2 && J.dz1 && J.grads, SMX_E_NULL);
        SMX_REQUIRE(J.rows > 0 && J.rows < (1 << 30), SMX_E_SHAPE);
    }
    // dz2 = (dz3 . W3) * relu'(h2)   [R, H2], K = OUT ;  dz1 = (dz2 . W2) * relu'(h1)   [R, H1], K = H2
    for (int stage = 0; stage < 2; ++stage) {
        GemmBatch G;
        G.n = njobs;
        int base = 0;
        for (int j = 0; j < njobs; ++j) {
            const smx_mlp3_job_t& J = jobs[j];
            const smx_mlp3_t* n = J.net;
            const int R = (int)J.rows;
            if (stage == 0)
                fill_prob(G.p[j], J.dz3, n->OUT, 1, n->W3, n->H2, 0, nullptr, J.h2, J.dz2, n->H2, R,
                          n->H2, n->OUT, SMX_ACT_NONE, nullptr, nullptr, base, J.stop_flag, J.dz2T, (int)(J.ldT ? J.ldT : R));
            else
                fill_prob(G.p[j], J.dz2, n->H2, 1, n->W2, n->H1, 0, nullptr, J.h1, J.dz1, n->H1, R,
                          n->H1, n->H2, SMX_ACT_NONE, nullptr, nullptr, base, J.stop_flag, J.dz1T, (int)(J.ldT ? J.ldT : R));
            base += G.p[j].tiles_m * G.p[j].tiles_n;
        }
        const int rc = launch_batch(G, smx_s(stream));
        if (rc) return rc;
    }
    return launch_wgrads(jobs, njobs, stream);
}

extern "C" int smx_mlp3_wgrad_multi_f32(const smx_mlp3_job_t* jobs, int32_t njobs, smx_stream_t stream) {
    SMX_REQUIRE(jobs, SMX_E_NULL);
    SMX_REQUIRE(njobs >= 1 && njobs <= MAX_JOBS, SMX_E_SHAPE);
    for (int j = 0; j < njobs; ++j) {
        const smx_mlp3_job_t& J = jobs[j];
        SMX_REQUIRE(J.net && J.grads && J.xT && J.h1T && J.h2T && J.dz1T && J.dz2T && J.dz3T, SMX_E_NULL);
        SMX_REQUIRE(J.rows > 0 && J.rows < (1 << 30), SMX_E_SHAPE);
    }
    return launch_wgrads(jobs, njobs, stream);
}

static int64_t mlp3_numel(const smx_mlp3_t* n);

extern "C" int smx_mlp3_backward_f32(const smx_mlp3_t* net, const float* x, const float* h1,
                                     const float* h2, const float* dz3, int64_t rows, float* dz2,
                                     float* dz1, float* grads, float* sumsq_partials,
                                     const int32_t* stop_flag, smx_stream_t stream) {
    smx_mlp3_job_t J;
    memset(&J, 0, sizeof(J));
    J.net = net; J.x = x; J.rows = rows; J.h1 = (float*)h1; J.h2 = (float*)h2; J.dz3 = dz3;
    J.dz2 = dz2; J.dz1 = dz1; J.grads = grads; J.sumsq_partials = sumsq_partials;
    J.stop_flag = stop_flag;
    SMX_REQUIRE(net, SMX_E_NULL);
    return smx_mlp3_backward_multi_f32(&J, 1, stream);
}

// The same over MANY rows (the stems' MLPs: rows = B x T ~ 10^4 - 10^5).  smx_mlp3_backward_f32 gives each 32 x 32 tile
// of a weight gradient to ONE workgroup that walks every row -- 117 workgroups at [300, 200] hidden sizes, 58 us at
// 7936 rows, 0.9 ms at 127 k.  Here the rows are cut into S chunks ((tile, chunk) -> a workgroup, partial tiles in the
// caller's workspace) and ONE segmented reduce forms the six gradients.  No sum-of-squares partials (the stem path takes
// the norm of the whole parameter group afterwards).
static int mlp3_splits(const smx_mlp3_t* n, int64_t rows) {
    // the hidden layers' gradients on smx_wgrad.hip (one workgroup per CU holds a whole dW and walks ONE chunk of rows):
    // as many chunks as CUs, chunks of >= 128 rows
    int wm, wn;
    if (rows >= SMX_WGRAD_ROWS_MIN && n->D % 4 == 0 && n->H1 % 4 == 0 && n->H2 % 4 == 0 &&
        (smx_wgrad_rows_plan(n->H1, n->D, &wm, &wn) || smx_wgrad_rows_plan(n->H2, n->H1, &wm, &wn))) {
        const long s = rows / 128;
        return (int)(s > 256 ? 256 : s);
    }
    const int t = ((n->H1 + 31) / 32) * ((n->D + 31) / 32) + ((n->H2 + 31) / 32) * ((n->H1 + 31) / 32) +
                  ((n->OUT + 31) / 32) * ((n->H2 + 31) / 32);
    long s = rows / 1024;
    const long cap = 4096 / t > 1 ? 4096 / t : 1;
    if (s > cap) s = cap;
    if (s > 64) s = 64;
    return s < 2 ? 1 : (int)s;
}
static int64_t mlp3_numel(const smx_mlp3_t* n) {
    return (int64_t)n->H1 * n->D + n->H1 + (int64_t)n->H2 * n->H1 + n->H2 + (int64_t)n->OUT * n->H2 + n->OUT;
}

extern "C" int64_t smx_mlp3_backward_ws_floats(int32_t D, int32_t H1, int32_t H2, int32_t OUT, int64_t rows) {
    smx_mlp3_t n;
    memset(&n, 0, sizeof(n));
    n.D = D; n.H1 = H1; n.H2 = H2; n.OUT = OUT;
    if (D <= 0 || H1 <= 0 || H2 <= 0 || OUT <= 0 || rows <= 0) return 0;
    const int S = mlp3_splits(&n, rows);
    return S > 1 ? (int64_t)S * mlp3_numel(&n) : 0;
}

// the weight-gradient half of the many-row backward: split-K partials in the workspace + one segmented reduce
static int mlp3_wgrads_splitk(const smx_mlp3_t* net, const float* x, const float* h1, const float* h2, const float* dz3,
                              const int R, const float* dz2, const float* dz1, float* grads, float* ws, int S,
                              const int32_t* stop_flag, smx_stream_t stream) {
    const int D = net->D, H1 = net->H1, H2 = net->H2, O = net->OUT;
    int k_chunk = ((R + S - 1) / S + 31) & ~31;
    while ((long)(S - 1) * k_chunk >= R) --S;                 // every chunk non-empty
    const float* dz[3] = {dz1, dz2, dz3};
    const float* in[3] = {x, h1, h2};
    const int Ms[3] = {H1, H2, O}, Ns[3] = {D, H1, H2};
    float* gdst = grads;
    float* wsp = ws;
    // three launches at most: the wide layers straight from the row-major operands (smx_wgrad.hip: the whole dW in one
    // workgroup's registers; SMX_WGRAD_TILED=1 keeps them on the tiled GEMM for A/B runs), layers wide enough for
    // the 64 x 64 tile kernel but not for that one, and the rest (the output layer: 1 - 17 rows of dW3) on 32 x 32 tiles
    // -- launch_batch() takes a batch to one kernel as a whole
    static const bool tiled_only = getenv("SMX_WGRAD_TILED") != nullptr;
    GemmBatch Gw, Gr;
    WgradBatch Gd;
    Gw.n = Gr.n = Gd.n = 0;
    Gd.stop = (const int*)stop_flag;
    RedSegs L;
    L.n = 6;
    int base_w = 0, base_r = 0, ebase = 0;
    for (int l = 0; l < 3; ++l) {
        const int M = Ms[l], N = Ns[l];
        float* wpart = wsp;
        float* bpart = wsp + (size_t)S * M * N;
        wsp = bpart + (size_t)S * M;
        int gw;
        if (!tiled_only && smx_wgrad_rows_eligible(dz[l], M, in[l], N, M, N, R) && smx_wgrad_rows_groups(M, N, &gw) == 1) {
            WgradProb& P = Gd.p[Gd.n++];
            P.A = dz[l]; P.B = in[l]; P.Cpart = wpart; P.bpart = bpart;
            P.M = M; P.N = N; P.lda = M; P.ldb = N; P.rows = R;
            P.ldc = N; P.c_split = (long)M * N; P.b_col0 = 0;
            P.splits = S; P.k_chunk = k_chunk;
        } else {
            const bool wide = M >= 48 && N >= 64 && M % 4 == 0 && N % 4 == 0;
            GemmBatch& G = wide ? Gw : Gr;
            int& base = wide ? base_w : base_r;
            GemmProb& P = G.p[G.n++];
            fill_prob(P, dz[l], M, 0, in[l], N, 0, nullptr, nullptr, wpart, N, M, N, R, SMX_ACT_NONE, bpart, nullptr, base,
                      stop_flag);
            P.splits = S;
            P.k_chunk = k_chunk;
            P.c_split = (long)M * N;
            base += P.tiles_m * P.tiles_n * S;
        }
        L.g[2 * l] = RedSeg{wpart, gdst, ebase, M * N};
        ebase += M * N;
        gdst += (size_t)M * N;
        L.g[2 * l + 1] = RedSeg{bpart, gdst, ebase, M};
        ebase += M;
        gdst += M;
    }
    L.total = ebase;
    if (Gd.n) {
        const int rc = smx_wgrad_rows_launch(Gd, smx_s(stream));
        if (rc) return rc;
    }
    // both on the 32 x 32 kernel (the wide layers' batch below the tile kernel's break-even: B*E ~ 10^4 rows): ONE launch --
    // the output layer's 12 us launch ran behind the hidden layers' instead of beside them
    if (Gw.n && Gr.n && Gw.n + Gr.n <= MAX_PROBS && !wgrad_batch_takes_tiles(Gw)) {
        batch_append(Gw, Gr);
        Gr.n = 0;
    }
    for (GemmBatch* G : {&Gw, &Gr}) {
        if (!G->n) continue;
        const int rc = launch_batch(*G, smx_s(stream));
        if (rc) return rc;
    }
    hipLaunchKernelGGL(segmented_reduce_kernel, dim3((unsigned)((L.total + 15) / 16)), dim3(256), 0, smx_s(stream), L, S,
                       stop_flag);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_mlp3_backward_splitk_f32(const smx_mlp3_t* net, const float* x, const float* h1, const float* h2,
                                            const float* dz3, int64_t rows, float* dz2, float* dz1, float* grads,
                                            float* ws, int64_t ws_floats, const int32_t* stop_flag,
                                            smx_stream_t stream) {
    SMX_REQUIRE(net && x && h1 && h2 && dz3 && dz2 && dz1 && grads, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && rows < (1 << 30), SMX_E_SHAPE);
    int S = mlp3_splits(net, rows);
    if (S <= 1 || !ws || ws_floats < (int64_t)S * mlp3_numel(net))
        return smx_mlp3_backward_f32(net, x, h1, h2, dz3, rows, dz2, dz1, grads, nullptr, stop_flag, stream);
    const int R = (int)rows, H1 = net->H1, H2 = net->H2, O = net->OUT;
    // dz2 = (dz3 . W3) * relu'(h2), dz1 = (dz2 . W2) * relu'(h1): as smx_mlp3_backward_multi_f32
    for (int stage = 0; stage < 2; ++stage) {
        GemmBatch G;
        G.n = 1;
        if (stage == 0)
            fill_prob(G.p[0], dz3, O, 1, net->W3, H2, 0, nullptr, h2, dz2, H2, R, H2, O, SMX_ACT_NONE, nullptr, nullptr, 0,
                      stop_flag);
        else
            fill_prob(G.p[0], dz2, H2, 1, net->W2, H1, 0, nullptr, h1, dz1, H1, R, H1, H2, SMX_ACT_NONE, nullptr, nullptr,
                      0, stop_flag);
        const int rc = launch_batch(G, smx_s(stream));
        if (rc) return rc;
    }
    return mlp3_wgrads_splitk(net, x, h1, h2, dz3, R, dz2, dz1, grads, ws, S, stop_flag, stream);
}

// The same with the data gradients as ONE fused launch (smx_mlp3_bwd16.hip: dz3 -> dz2 -> dz1 -> dx with the
// intermediates handed on in registers) in front of the split-K weight gradients.  dx [rows, D] (may be NULL): the
// gradient with respect to the MLP's input, which the layered path leaves to a separate smx_linear_f32 call.
// packedT: scratch of smx_mlp3_dgrad_rows_ws_floats() floats for the transposed packed weights.
// SMX_E_UNSUPPORTED (nothing launched) outside the fused kernel's shapes or without a split-K workspace: the caller
// then uses smx_mlp3_backward_splitk_f32 + smx_linear_f32.
extern "C" int smx_mlp3_backward_rows_f32(const smx_mlp3_t* net, const float* x, const float* h1, const float* h2,
                                          const float* dz3, int64_t rows, float* dz2, float* dz1, float* dx, float* grads,
                                          float* ws, int64_t ws_floats, float* packedT, int64_t packedT_floats,
                                          const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(net && x && h1 && h2 && dz3 && dz2 && dz1 && grads && packedT, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && rows < (1 << 30), SMX_E_SHAPE);
    const int S = mlp3_splits(net, rows);
    if (S <= 1 || !ws || ws_floats < (int64_t)S * mlp3_numel(net)) return SMX_E_UNSUPPORTED;
    const int rc = smx_mlp3_dgrad_rows_launch(net, h1, h2, dz3, rows, dz2, dz1, dx, packedT, packedT_floats, stop_flag,
                                              smx_s(stream));
    if (rc) return rc;
    return mlp3_wgrads_splitk(net, x, h1, h2, dz3, (int)rows, dz2, dz1, grads, ws, S, stop_flag, stream);
}

