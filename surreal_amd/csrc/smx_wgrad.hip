// Weight gradients over MANY rows: dW[M, N] = dZ^T . X with dZ [rows, M], X [rows, N] row-major and rows ~ 10^5
// (loss.backward() through the nn.Linear layers of the MLPs on top of an LSTM / CNN stem, surreal/learner/ppo.py:227-353
// over B*E rows; the reference leaves this product to ATen).
//
// Why its own kernel.  In this product the REDUCTION index is the row, i.e. the slow dimension of both operands.
//   * The tiled GEMM (gemm_tile_kernel<2, 2>) stages 32-row blocks in LDS as [k][rows + 4], reads every fragment word with
//     its own ds_read_b32, and gives every 64 x 64 tile of dW its own walk over all rows: 412 us for the two hidden layers'
//     gradients at 126 976 rows (23 GFLOP: 55 TFLOP/s, profiles/r05_lstm_1024x128_kernel_stats.csv).
//   * For v_mfma_f32_16x16x4 "K-strided" is the natural operand order: lane (i = lane & 15, kq = lane >> 4) holds
//     A(k = kq, m = i) and B(k = kq, n = i), so a fragment of 4 rows x 16 columns is four 64-byte row segments -- and with
//     ONE w-dword load per lane (columns w i .. w i + w - 1 of row kq) a wavefront holds the fragments of w tiles whose
//     columns interleave (tile j = columns w i + j): a permutation of the tile's columns that is undone when the
//     accumulators are stored.  No LDS, no barrier: per 4 rows a wavefront issues 2 - 4 loads of whole row segments.
//   * A first version of this kernel (wave tiles of 5 x 5 tiles, every wave tile walking the rows by itself) was as slow as
//     the tiled GEMM (382 against 412 us): 1.4 GB of operand reads per launch -- every row once per wave-tile column --
//     against 456 MB of operands; the tiles of a chunk drift apart and their re-reads miss the 4 MB L2.  So: the WHOLE dW
//     lives in the registers of ONE workgroup (200 x 300 words = 240 KB of the CU's 512 KB register file: eight
//     wavefronts, 2 x 4 or 4 x 2, each with up to 7 x 5 tiles = 140 accumulator registers), every operand row is read from
//     memory once, and only the rows are split over the workgroups (one per CU; partial matrices in the caller's
//     workspace, a segmented reduce adds them in split order: deterministic).  Bias gradients (column sums of dZ) ride on
//     the A fragments of the wavefronts in the first column.
#include "smx_common.h"
#include "smx_wgrad.h"

namespace {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    const uintptr_t u = (uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void* q = (void*)(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// W consecutive floats (W = 0: nothing).  Words past the buffer's end come back as 0; words past the matrix's last column
// but inside the buffer are the next row's first words: they only ever meet accumulators of columns that are not stored.
// (exactly W registers: a float4 for every word cost the ring a fourth of its depth)
template <int W> struct Words { float v[W > 0 ? W : 1]; };
template <int W>
__device__ __forceinline__ Words<W> ldw(rsrc_t R, unsigned off) {
    Words<W> o;
    if (W == 0) {
        o.v[0] = 0.f;
    } else if (W == 1) {
        o.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R, off, 0, 0));
    } else if (W == 2) {
        const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(R, off, 0, 0);
        o.v[0] = __uint_as_float(w.x); o.v[1 % (W > 0 ? W : 1)] = __uint_as_float(w.y);
    } else if (W == 3) {
        const u32x3 w = __builtin_amdgcn_raw_buffer_load_b96(R, off, 0, 0);
        o.v[0] = __uint_as_float(w.x); o.v[1 % (W > 0 ? W : 1)] = __uint_as_float(w.y); o.v[2 % (W > 0 ? W : 1)] = __uint_as_float(w.z);
    } else {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(R, off, 0, 0);
        o.v[0] = __uint_as_float(w.x); o.v[1 % (W > 0 ? W : 1)] = __uint_as_float(w.y); o.v[2 % (W > 0 ? W : 1)] = __uint_as_float(w.z);
        o.v[3 % (W > 0 ? W : 1)] = __uint_as_float(w.w);
    }
    return o;
}

// One wavefront: MT x NT tiles of 16 x 16.  Along each dimension the T tiles are two load groups: G0 = min(T, 4) tiles
// from one G0-dword load (tile j = columns c0 + G0 i + j), G1 = T - G0 tiles from a G1-dword load
// (tile G0 + j = columns c0 + 16 G0 + G1 i + j).
template <int MT, int NT>
__device__ __attribute__((noinline)) void wgrad_wave(const WgradProb& P, const int m0, const int n0, const int r_lo,
                                                     const int r_hi, const bool bias) {
    constexpr int MA = MT < 4 ? MT : 4, MB = MT - MA, NA = NT < 4 ? NT : 4, NB = NT - NA;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const rsrc_t rA = make_rsrc(P.A, (unsigned)P.rows * (unsigned)P.lda * 4u);
    const rsrc_t rB = make_rsrc(P.B, ((unsigned)P.rows * (unsigned)P.ldb - (unsigned)P.b_col0) * 4u);
    // byte offsets of this lane's words in row kq of a step; OOB when the lane's first column is past the matrix
    const unsigned a0 = (m0 + MA * i < P.M) ? ((unsigned)kq * P.lda + m0 + MA * i) * 4u : OOB;
    const unsigned a1 = (MB > 0 && m0 + 16 * MA + MB * i < P.M) ? ((unsigned)kq * P.lda + m0 + 16 * MA + MB * i) * 4u : OOB;
    const unsigned b0 = (n0 + NA * i < P.N) ? ((unsigned)kq * P.ldb + n0 + NA * i) * 4u : OOB;
    const unsigned b1 = (NB > 0 && n0 + 16 * NA + NB * i < P.N) ? ((unsigned)kq * P.ldb + n0 + 16 * NA + NB * i) * 4u : OOB;
    const unsigned sa = (unsigned)P.lda * 4u, sb = (unsigned)P.ldb * 4u;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int jm = 0; jm < MT; ++jm)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[jm][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs[MT];
#pragma unroll
    for (int jm = 0; jm < MT; ++jm) bs[jm] = 0.f;

    // the ring of operand registers: FOUR steps (of 4 rows) in flight ahead of the MFMAs -- 3 x 2000 cycles between a
    // load and its use: a CU asks memory for 8 KB per step -- as NAMED variables.
    // Rows at or past r_hi: every offset out of range (no memory traffic, zeros: the MFMAs add nothing).
    Words<MA> pa0, pa1, pa2, pa3;
    Words<MB> qa0, qa1, qa2, qa3;
    Words<NA> pb0, pb1, pb2, pb3;
    Words<NB> qb0, qb1, qb2, qb3;
#define WG_LOAD(PA, QA, PB, QB, k)                                                                           \
    do {                                                                                                     \
        const bool in_ = (k) + kq < r_hi;                                                                    \
        const unsigned ka = (unsigned)(k) * sa, kb = (unsigned)(k) * sb;                                     \
        const unsigned oa0 = (in_ && a0 != OOB) ? a0 + ka : OOB, ob0 = (in_ && b0 != OOB) ? b0 + kb : OOB;   \
        const unsigned oa1 = (in_ && a1 != OOB) ? a1 + ka : OOB, ob1 = (in_ && b1 != OOB) ? b1 + kb : OOB;   \
        /* the issue ORDER is pinned: s_waitcnt vmcnt counts back from the newest load, and where the */     \
        /* prologue and the loop body disagree about it hipcc waits for vmcnt(0) at the loop's top */        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        PA = ldw<MA>(rA, oa0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        PB = ldw<NA>(rB, ob0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        QA = ldw<MB>(rA, oa1);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        QB = ldw<NB>(rB, ob1);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define WG_STEP(PA, QA, PB, QB, knext)                                                                       \
    do {                                                                                                     \
        float av[MT], bv[NT];                                                                                \
        _Pragma("unroll") for (int jm = 0; jm < MT; ++jm) av[jm] = jm < MA ? PA.v[jm < MA ? jm : 0] : QA.v[jm >= MA ? jm - MA : 0]; \
        _Pragma("unroll") for (int jn = 0; jn < NT; ++jn) bv[jn] = jn < NA ? PB.v[jn < NA ? jn : 0] : QB.v[jn >= NA ? jn - NA : 0]; \
        _Pragma("unroll") for (int jm = 0; jm < MT; ++jm)                                                    \
            _Pragma("unroll") for (int jn = 0; jn < NT; ++jn) acc[jm][jn] = MFMA16(av[jm], bv[jn], acc[jm][jn]); \
        if (bias) {                                                                                          \
            _Pragma("unroll") for (int jm = 0; jm < MT; ++jm) bs[jm] += av[jm];                              \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        WG_LOAD(PA, QA, PB, QB, knext);                                                                      \
    } while (0)
    WG_LOAD(pa0, qa0, pb0, qb0, r_lo);
    WG_LOAD(pa1, qa1, pb1, qb1, r_lo + 4);
    WG_LOAD(pa2, qa2, pb2, qb2, r_lo + 8);
    WG_LOAD(pa3, qa3, pb3, qb3, r_lo + 12);
    for (int k = r_lo; k < r_hi; k += 16) {
        WG_STEP(pa0, qa0, pb0, qb0, k + 16);
        WG_STEP(pa1, qa1, pb1, qb1, k + 20);
        WG_STEP(pa2, qa2, pb2, qb2, k + 24);
        WG_STEP(pa3, qa3, pb3, qb3, k + 28);
    }
#undef WG_STEP
#undef WG_LOAD

    // ---- store: C-fragment register r of tile (jm, jn) is row-index 4 kq + r, column-index i of the tile ------------
    float* C = P.Cpart;                                 // (the kernel has moved it to this split's partial matrix)
#pragma unroll
    for (int jm = 0; jm < MT; ++jm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (jm < MA) ? m0 + MA * (4 * kq + r) + jm : m0 + 16 * MA + MB * (4 * kq + r) + (jm - MA);
            if (m < P.M) {
                float* row = C + (size_t)m * P.ldc;
                const int c0 = n0 + NA * i, c1 = n0 + 16 * NA + NB * i;
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    const int n = (jn < NA) ? c0 + jn : c1 + (jn - NA);
                    if (n < P.N) row[n] = acc[jm][jn][r];
                }
            }
        }
    if (bias) {
        // column sums of dZ: over the four row groups of the fragment
#pragma unroll
        for (int jm = 0; jm < MT; ++jm) {
            bs[jm] += __shfl_xor(bs[jm], 16, 64);
            bs[jm] += __shfl_xor(bs[jm], 32, 64);
        }
        if (kq == 0) {
#pragma unroll
            for (int jm = 0; jm < MT; ++jm) {
                const int m = (jm < MA) ? m0 + MA * i + jm : m0 + 16 * MA + MB * i + (jm - MA);
                if (m < P.M) P.bpart[m] = bs[jm];
            }
        }
    }
}

// tiles [first, first + count) of a dimension of T tiles cut over `parts` wavefronts: the first T % parts get one more
__device__ __forceinline__ void cut(int T, int parts, int idx, int& first, int& count) {
    const int q = T / parts, rem = T - q * parts;
    count = q + (idx < rem ? 1 : 0);
    first = q * idx + (idx < rem ? idx : rem);
}

__global__ __launch_bounds__(512, 2) void wgrad_rows_kernel(WgradBatch G) {
    if (G.stop && *G.stop) return;
    const int b = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int k = 1; k < WGRAD_MAX_PROBS; ++k) pi += (k < G.n && b >= G.p[k].wg_base) ? 1 : 0;
    WgradProb P = G.p[pi];
    const int s = b - P.wg_base;                              // the split (chunk of rows) of this workgroup
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wv / P.wn, wj = wv - wi * P.wn;            // this wavefront's block of tiles
    int fm, cm, fn, cn;
    cut((P.M + 15) >> 4, P.wm, wi, fm, cm);
    cut((P.N + 15) >> 4, P.wn, wj, fn, cn);
    const int r_lo = s * P.k_chunk;
    const int r_hi = (r_lo + P.k_chunk < P.rows) ? r_lo + P.k_chunk : P.rows;
    P.Cpart += (size_t)s * P.c_split;
    if (P.bpart) P.bpart += (size_t)s * P.M;
    const bool bias = wj == 0 && P.bpart != nullptr;
    const int m0 = 16 * fm, n0 = 16 * fn;
#define WG_CASE(MT, NT) else if (cm == MT && cn == NT) wgrad_wave<MT, NT>(P, m0, n0, r_lo, r_hi, bias)
    if (false) {}
    WG_CASE(7, 5); WG_CASE(7, 4); WG_CASE(7, 3);
    WG_CASE(6, 5); WG_CASE(6, 4); WG_CASE(6, 3);
    WG_CASE(5, 5); WG_CASE(5, 4); WG_CASE(5, 3);
    WG_CASE(4, 5); WG_CASE(4, 4); WG_CASE(4, 3);
#undef WG_CASE
}

// wave-tile sizes the kernel is instantiated for
inline bool size_ok(int T, int parts, int lo, int hi) {
    const int q = T / parts, rem = T - q * parts;
    const int big = q + (rem ? 1 : 0), small = q;
    return small >= lo && big <= hi;
}

}  // namespace

// -> the 8 wavefronts as wm x wn over the (M / 16) x (N / 16) tiles, or false when no cut fits the instantiated sizes
// (MT in 4..7, NT in 3..5: dW of up to 448 x 320)
bool smx_wgrad_rows_plan(int M, int N, int* wm, int* wn) {
    const int Tm = (M + 15) / 16, Tn = (N + 15) / 16;
    const int cand[4][2] = {{2, 4}, {4, 2}, {8, 1}, {1, 8}};
    int best = -1, bestcost = 1 << 30;
    for (int c = 0; c < 4; ++c) {
        const int a = cand[c][0], b = cand[c][1];
        if (!size_ok(Tm, a, 4, 7) || !size_ok(Tn, b, 3, 5)) continue;
        const int cost = ((Tm + a - 1) / a) * ((Tn + b - 1) / b);         // the busiest wavefront's MFMAs per step
        if (cost < bestcost) { best = c; bestcost = cost; }
    }
    if (best < 0) return false;
    *wm = cand[best][0]; *wn = cand[best][1];
    return true;
}

int smx_wgrad_rows_groups(int M, int N, int* width) {
    int wm, wn;
    if (smx_wgrad_rows_plan(M, N, &wm, &wn)) { *width = N; return 1; }
    for (int g = 2; g <= WGRAD_MAX_PROBS; ++g) {
        const int w = (((N + g - 1) / g) + 15) & ~15;          // whole 16-column tiles per group
        const int last = N - (g - 1) * w;
        if (last <= 0) continue;
        if (smx_wgrad_rows_plan(M, w, &wm, &wn) && smx_wgrad_rows_plan(M, last, &wm, &wn)) { *width = w; return g; }
    }
    return 0;
}

bool smx_wgrad_rows_eligible(const float* A, int lda, const float* B, int ldb, int M, int N, long rows) {
    int width;
    return rows >= SMX_WGRAD_ROWS_MIN && M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
           ((((uintptr_t)A | (uintptr_t)B) & 15) == 0) && rows * (long)lda * 4 < (1l << 31) &&
           rows * (long)ldb * 4 < (1l << 31) && smx_wgrad_rows_groups(M, N, &width) > 0;
}

int smx_wgrad_rows_launch(WgradBatch& G, hipStream_t st) {
    int base = 0;
    for (int k = 0; k < G.n; ++k) {
        WgradProb& P = G.p[k];
        if (!smx_wgrad_rows_plan(P.M, P.N, &P.wm, &P.wn)) return SMX_E_UNSUPPORTED;
        P.wg_base = base;
        base += P.splits;
    }
    hipLaunchKernelGGL(wgrad_rows_kernel, dim3((unsigned)base), dim3(512), 0, st, G);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}
