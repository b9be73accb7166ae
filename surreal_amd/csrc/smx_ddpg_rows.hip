// One DDPG iteration on ROW BLOCKS (surreal/learner/ddpg.py:244-352; low-dimensional observations, one critic):
// the layer-by-layer schedule is ~19 dependent launches of 512-row problems, each a few microseconds of work behind a
// launch boundary and an L2 first touch (0.19 ms per iteration, 70 % of it in 15 dense launches).  Batch rows are
// independent up to the weight gradients, so here a workgroup owns FOUR rows and carries them through whole chains on the
// 4-row loop of the rollout kernel (smx_rows4_mma.inc.h: v_mfma_f32_4x4x1, weights in the fragment order of
// smx_epoch_pack.inc.h) -- configs[2]'s batch of 512 is 128 workgroups:
//
//   ddpg_rows4_kernel<0>      target actor -> target critic -> Q'(s', mu'(s'));  critic -> Q(s, a);  y and dLoss/dQ;
//                             the critic's data gradients dz2, dz1;  the actor's forward pass for ITS update (it reads
//                             only the actor's parameters, which the critic update does not touch)
//   ddpg_rows_wgrad_update    the critic's weight gradients, Adam, its target network's update, the packed copies
//   ddpg_rows4_kernel<1>      Q(s, mu(s)) through the UPDATED critic; d(-mean Q)/d(action); tanh'; the actor's data
//                             gradients dz2, dz1
//   ddpg_rows_wgrad_update    the same for the actor;  [statistics: smx_ddpg_stats_f32]
//
// Activations and gradients that the weight-gradient launches read go to HBM row-major, exactly the buffers of the
// layer-by-layer schedule.  Products are summed in the MFMA loop's order (32-wide K chunks, k ascending per lane group),
// not in smx_linear_f32's: results agree with the layered schedule to fp32 rounding, not bit for bit.
//
// History: round 5 built this on 16-row blocks and one noinline dense function (32 workgroups bound by the matrix pipes of
// 32 CUs, 92 + 46 us for the two chains against the 145 us of the launches they replaced: slower, off by default); round 6
// moved it to 4-row blocks (bound by what a CU pulls from L2 -- every workgroup streams every weight once per layer -- on
// four times as many CUs: 42 + 22 us) and removed the 16-row kernels: they were never faster than the level schedule.
#include "smx_common.h"
#include <string.h>

#define SMX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace {
#include "smx_epoch_pack.inc.h"
#include "smx_epoch_mma.inc.h"
#include "smx_rows4_mma.inc.h"
#include "smx_ddpg_stats.inc.h"

constexpr int DNWV = 8;           // wavefronts per workgroup: two per SIMD (one's loads hide under the other's MFMAs)
constexpr int DNTH = 64 * DNWV;
constexpr int RBLK = 4;           // batch rows per workgroup
constexpr int LDO = 36;           // row stride of an output tile (<= 32 outputs)
constexpr int LDK4 = 80;          // row stride of a tile that is the K <= 32 input of a product (two zero-padded chunks; strides = 16 mod 64: the 16 (row, kq) readers on distinct banks)
constexpr int DTG4 = 4;           // feature tiles a wave carries per pass (8 x 4 = 32 tiles: 400 features in one pass)
constexpr int MAX_LDS = 128 * 1024;
enum { A_NONE = 0, A_RELU = 1, A_TANH = 2, A_MASK = 3 };

__host__ __device__ inline int r64(int v) { return (v + 63) & ~63; }

struct PMat {                     // a matrix in fragment order (smx_epoch_pack.inc.h): M features x K inputs
    const float* P;
    int M, K;
};

// Phase timestamps (cycle counter of thread 0 of every workgroup into a caller-supplied buffer, 128 slots per workgroup:
// 0 .. 15 the launch's phases, 16 + 5 k .. the k-th layer's entry / K loop / epilogue / barrier) exist only in a build with
// -DSMX_DDPG_TIMING (scripts/bench_ddpg_rows.py); the product build has none.
#ifdef SMX_DDPG_TIMING
#define TSTAMP(i) do { if (G.tbuf && threadIdx.x == 0) G.tbuf[(size_t)blockIdx.x * 128 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

struct RNet {                     // a network's biases (row-major parameter buffer) and packed weights
    const float *b1, *b2, *b3;
    PMat W1, W2, W3;
};

struct RArgs {
    int rows, D, A, H1, H2, c1, c2;
    RNet a, c, ta, tc;
    PMat cW2Tlo, cW2Thi, aW3T, aW2T;      // the transposed blocks of the backward products
    const float* cW3;                     // the critic's output layer, row-major [c2]
    const float *x, *xn, *actions, *rewards, *dones;
    float gamma_n;
    float *xcat, *h2c, *q, *q_next, *y, *dz3, *dz2, *dxcat;         // critic phase (xcat / dxcat: row stride c1 + A)
    float *h1a, *h2a, *act;                                          // the actor's forward pass
    float *q_actor, *dz3a, *dz2a, *dz1a;                             // actor phase
    int* step;
    // LDS carve-up (float offsets)
    int ldx, ldA, ldB, ldC, oX, oXn, oA, oB, oC, oO, oO2, oO3, oZ, oS, oR, total;
    long long* tbuf;              // SMX_DDPG_TIMING builds
};

__device__ __forceinline__ void zero_lds(int total) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < (total >> 2); i += DNTH) *(float4*)(sm + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// rows row0 .. row0 + 3 of a row-major [rows][D] matrix -> an LDS tile (zero past the batch)
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int ld_src, int cols, long row0, int nrows,
                                           int off, int ld) {
    extern __shared__ float sm[];
    for (int idx = threadIdx.x; idx < RBLK * cols; idx += DNTH) {
        const int n = idx / cols, j = idx - n * cols;
        sm[off + n * ld + j] = (n < nrows) ? src[(size_t)(row0 + n) * ld_src + j] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The two chains as a LAYER PROGRAM.  With one noinline dense function called a dozen times (round 5's structure) the
// calls were the larger half of a launch whose K loops are 2 - 9 k cycles (phase stamps: 3.3 k cycles per layer between
// return and the next entry -- callee-saved registers through scratch -- and 1.1 k from entry to the first product).
// Here the host writes the chain as a table of steps in the kernel arguments and the
// kernel is ONE loop over it: the layer body exists once, inline, its operands arrive by scalar loads.  What sits between
// two layers of a chain (the concat of the action, the Bellman target and dLoss/dQ, tanh') is the `post` of the step
// before it.
//
//  * lane (fm, kq) keeps row kq of feature 16 t + fm: the kq groups meet by meet_rows (3 swaps, 3 adds)
//  * layers of at most two feature tiles (the heads: 6 actions, 1 value; the action gradient) split K over the eight
//    wavefronts instead of leaving 64 - 80 dependent products to one: a wave takes ceil(C2 / 8) chunks, the eight partial
//    sums meet through LDS in wave order (k ascending inside a wave's chunks: a fixed order, not the full-K order)
// ---------------------------------------------------------------------------------------------------------------
enum { P_NONE = 0, P_TC_CAT, P_C_CAT, P_LOSS, P_A_CAT, P_A_DQ, P_A_TANH };
constexpr int MAX_STEPS = 13;

struct Step {
    const float* W;               // packed weights
    const float* bias;            // [M] or null
    float* g;                     // HBM output [rows][ldg] or null
    int M, K, in_off, ldi, act, mask_off, ldm, out_off, ldo, ldg, post;
};
struct Prog {
    int n;
    Step s[MAX_STEPS];
};

#ifdef SMX_DDPG_TIMING
#define PSTAMP(k, i) do { if (G.tbuf && threadIdx.x == 0) G.tbuf[(size_t)blockIdx.x * 128 + 16 + 5 * (k) + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PSTAMP(k, i) do { } while (0)
#endif

template <int PHASE>
__global__ __launch_bounds__(DNTH) void ddpg_rows4_kernel(RArgs G, Prog P) {
    extern __shared__ float sm[];
    constexpr int RB = RBLK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 15, kq = lane >> 4;
    const long row0 = (long)blockIdx.x * RB;
    int nrows = G.rows - (int)row0;
    nrows = nrows > RB ? RB : nrows;
    const int A = G.A, c1 = G.c1, c2 = G.c2, ldc = c1 + A;
    TSTAMP(0);
    zero_lds(G.total);
    __syncthreads();
    stage_rows(G.x, G.D, G.D, row0, nrows, G.oX, G.ldx);
    float rew = 0.f, dn = 0.f;
    if (PHASE == 0) {
        stage_rows(G.xn, G.D, G.D, row0, nrows, G.oXn, G.ldx);
        if (tid < nrows) { rew = G.rewards[row0 + tid]; dn = G.dones[row0 + tid]; }
    } else {
        stage_rows(G.act, A, A, row0, nrows, G.oO, LDO);
    }
    SMX_LDS_BARRIER();

#pragma unroll 1
    for (int si = 0; si < P.n; ++si) {
        const Step& S = P.s[si];
        const int M = S.M, K = S.K, act = S.act, ldi = S.ldi, ldm = S.ldm, ldo = S.ldo, ldg = S.ldg;
        const int mask_off = S.mask_off, out_off = S.out_off;
        const int tiles = (M + 15) >> 4, C2 = pack_chunks(K);
        const rsrc_t rw = make_rsrc(S.W, (unsigned)tiles * (unsigned)C2 * 2048u);
        const rsrc_t rb = make_rsrc(S.bias ? S.bias : S.W, S.bias ? (unsigned)M * 4u : 0u);
        const rsrc_t rg = make_rsrc(S.g ? S.g : S.W, S.g ? (unsigned)G.rows * (unsigned)ldg * 4u : 0u);
        const float* in = sm + S.in_off;
        PSTAMP(si, 0);
        // row `r` of feature f is finished: bias, activation, the tile for the next layer, the row-major copy
        auto finish = [&](float z, float bias_v, int f, int r) {
            z += bias_v;
            if (act == A_RELU) z = (z < 0.f) ? 0.f : z;
            else if (act == A_TANH) z = tanhf(z);
            else if (act == A_MASK) z = (sm[mask_off + r * ldm + f] > 0.f) ? z : 0.f;
            const float v = (f < M) ? z : 0.f;
            if (out_off >= 0) sm[out_off + r * ldo + f] = v;
            const bool ok = r < nrows && f < M;                      // (no HBM output: every offset is out of range)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rg,
                                                  ok ? ((unsigned)(row0 + r) * (unsigned)ldg + (unsigned)f) * 4u : OOB, 0, 0);
        };
        if (tiles <= 2) {
            // ---- split K: wave w takes chunks [w per, w per + per) of both tiles ----
            const int per = (C2 + DNWV - 1) / DNWV;
            const int ca = wv * per;
            int cb = ca + per;
            cb = cb > C2 ? C2 : cb;
            const int ft = 16 * (tid >> 6) + fm;                     // the (tile, row, feature) thread tid finishes
            const float bfin = ld4(rb, (tid < 64 * tiles && ft < M) ? (unsigned)ft * 4u : OOB);
            f32x4 a2[2][1];
            a2[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            a2[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* bp = in + (lane & 3) * ldi + 8 * kq;
            unsigned wo[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) wo[g] = (g < tiles) ? ((unsigned)g * (unsigned)C2 * 512u + (unsigned)lane * 4u) * 4u : OOB;
#pragma unroll 1
            for (int c = ca; c < cb; c += 2) {
                const unsigned none[2] = {OOB, OOB};
                WFrag4<2, 1> F0, F1;
                ld_wfrag4<2, 1>(F0, rw, wo, bp, ldi, c);
                if (c + 1 < cb) ld_wfrag4<2, 1>(F1, rw, wo, bp, ldi, c + 1);
                else ld_wfrag4<2, 1>(F1, rw, none, bp, ldi, 0);
                mma4_chunk<2, 1>(a2, F0);
                mma4_chunk<2, 1>(a2, F1);
            }
            PSTAMP(si, 1);
            float* red = sm + G.oR;                                  // [wave][tile][row][16]
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (g < tiles) red[((wv * 2 + g) * 4 + kq) * 16 + fm] = meet_rows(a2[g][0]);
            SMX_LDS_BARRIER();
            PSTAMP(si, 2);
            if (tid < 64 * tiles) {
                const int g = tid >> 6;
                float z = red[(g * 4 + kq) * 16 + fm];
#pragma unroll
                for (int w = 1; w < DNWV; ++w) z += red[((w * 2 + g) * 4 + kq) * 16 + fm];
                finish(z, bfin, ft, kq);
            }
        } else {
#pragma unroll 1
            for (int tb = 0; tb < tiles; tb += DNWV * DTG4) {
                const int t0 = tb + wv;
                int nt = (tiles - t0 + DNWV - 1) / DNWV;
                nt = nt < 0 ? 0 : (nt > DTG4 ? DTG4 : nt);
                float bs[DTG4];
#pragma unroll
                for (int g = 0; g < DTG4; ++g) {
                    const int f = 16 * (t0 + DNWV * g) + fm;
                    bs[g] = ld4(rb, (g < nt && f < M) ? (unsigned)f * 4u : OOB);
                }
                f32x4 acc[DTG4];
#pragma unroll
                for (int g = 0; g < DTG4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define SMX_RUN4(N)                                                                  \
    {                                                                                \
        f32x4 a[N][1];                                                               \
        _Pragma("unroll") for (int g = 0; g < N; ++g) a[g][0] = acc[g];              \
        fwd_tiles4<N, 1>(a, rw, tiles, C2, in, ldi, t0, DNWV, lane);                 \
        _Pragma("unroll") for (int g = 0; g < N; ++g) acc[g] = a[g][0];              \
    }
                if (nt > 3) SMX_RUN4(4)
                else if (nt > 2) SMX_RUN4(3)
                else if (nt > 1) SMX_RUN4(2)
                else if (nt > 0) SMX_RUN4(1)
#undef SMX_RUN4
                if (tb == 0) PSTAMP(si, 1);
                float z[DTG4];
#pragma unroll
                for (int g = 0; g < DTG4; ++g) z[g] = meet_rows(acc[g]);
#pragma unroll
                for (int g = 0; g < DTG4; ++g)
                    if (g < nt) finish(z[g], bs[g], 16 * (t0 + DNWV * g) + fm, kq);      // (wave-uniform)
            }
            PSTAMP(si, 2);
        }
        PSTAMP(si, 3);
        SMX_LDS_BARRIER();
        PSTAMP(si, 4);

        // ---- what follows the layer in its chain ----
        const int post = S.post;
        if (PHASE == 0) {
            if (post == P_TC_CAT) {            // [h1' | mu'(s')]: the action behind the first c1 columns of the concat tile
                if (tid < RB * A) {
                    const int n = tid / A, j = tid - n * A;
                    sm[G.oC + n * G.ldC + c1 + j] = sm[G.oO + n * LDO + j];
                }
                SMX_LDS_BARRIER();
            } else if (post == P_C_CAT) {      // [h1 | a] and its row-major copy
                if (tid < RB * A) {
                    const int n = tid / A, j = tid - n * A;
                    const float v = (n < nrows) ? G.actions[(size_t)(row0 + n) * A + j] : 0.f;
                    sm[G.oC + n * G.ldC + c1 + j] = v;
                    if (n < nrows) G.xcat[(size_t)(row0 + n) * ldc + c1 + j] = v;
                }
                SMX_LDS_BARRIER();
            } else if (post == P_LOSS) {
                // y = r + gamma^n Q' (1 - done) (ddpg.py:279); dLoss/dQ of the mean squared error (ddpg.py:307-308)
                if (tid < RB) {
                    const float qn = sm[G.oO2 + tid * LDO], q = sm[G.oO + tid * LDO];
                    const float t = (G.gamma_n * qn) * (1.0f - dn);
                    const float yy = rew + t;
                    const float d3 = (2.0f * (q - yy)) / (float)G.rows;
                    sm[G.oS + tid] = (tid < nrows) ? d3 : 0.f;
                    if (tid < nrows) {
                        G.q[row0 + tid] = q;
                        G.q_next[row0 + tid] = qn;
                        G.y[row0 + tid] = yy;
                        G.dz3[row0 + tid] = d3;
                    }
                }
                if (blockIdx.x == 0 && tid == 0 && G.step) *G.step += 1;      // this iteration's Adam step (both groups)
                SMX_LDS_BARRIER();
                // dz2 = (dz3 W3) relu'(h2), a K = 1 product: elementwise
                for (int idx = tid; idx < RB * c2; idx += DNTH) {
                    const int n = idx / c2, j = idx - n * c2;
                    float v = sm[G.oS + n] * G.cW3[j];
                    v = (sm[G.oB + n * G.ldB + j] > 0.f) ? v : 0.f;
                    sm[G.oA + n * G.ldA + j] = v;
                    if (n < nrows) G.dz2[(size_t)(row0 + n) * c2 + j] = v;
                }
                SMX_LDS_BARRIER();
            }
        } else {
            if (post == P_A_CAT) {
                if (tid < RB * A) {
                    const int n = tid / A, j = tid - n * A;
                    sm[G.oC + n * G.ldC + c1 + j] = sm[G.oO + n * LDO + j];
                }
                SMX_LDS_BARRIER();
            } else if (post == P_A_DQ) {
                // the masks of the actor's backward pass (its forward pass ran in the critic phase): h1a -> the concat
                // tile, whose layer-2 product is done; h2a follows once dz2 has read the critic's ReLU mask
                stage_rows(G.h1a, G.H1, G.H1, row0, nrows, G.oC, G.ldC);
                const float dq = -1.0f / (float)G.rows;            // d(-mean Q)/d(h2) = (-1/rows) W3 relu'(h2)
                for (int idx = tid; idx < RB * c2; idx += DNTH) {
                    const int n = idx / c2, j = idx - n * c2;
                    float v = dq * G.cW3[j];
                    v = (n < nrows && sm[G.oB + n * G.ldB + j] > 0.f) ? v : 0.f;
                    sm[G.oA + n * G.ldA + j] = v;
                }
                SMX_LDS_BARRIER();
                stage_rows(G.h2a, G.H2, G.H2, row0, nrows, G.oB, G.ldB);
            } else if (post == P_A_TANH) {     // through tanh: the action gradient times 1 - a^2
                if (tid < RB * A) {
                    const int n = tid / A, j = tid - n * A;
                    const float a = sm[G.oO + n * LDO + j];
                    const float v = sm[G.oO2 + n * LDO + j] * (1.0f - a * a);
                    sm[G.oZ + n * LDK4 + j] = v;
                    if (n < nrows) G.dz3a[(size_t)(row0 + n) * A + j] = v;
                }
                SMX_LDS_BARRIER();
            }
        }
        TSTAMP(si + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the packed copies: one thread per 16-byte word of [tile][chunk][half][lane][4]
// ---------------------------------------------------------------------------------------------------------------
constexpr int N_MAT = 16;
struct PItem {
    const float* src;
    int ld, M, K, tr;             // X[m][k] = tr ? src[k ld + m] : src[m ld + k]
    long base;                    // first 16-byte word of the block in the packed buffer
};
struct PArgs {
    PItem it[N_MAT];
    float* packed;
    int count;
    long total;                   // words covered by the launch (items are contiguous from it[0].base)
};

__global__ __launch_bounds__(256) void ddpg_pack_kernel(PArgs P) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.total) return;
    const long w0 = i + P.it[0].base;
    int pi = 0;
#pragma unroll
    for (int k = 1; k < N_MAT; ++k) pi += (k < P.count && w0 >= P.it[k].base) ? 1 : 0;
    const PItem N = P.it[pi];
    const long w = w0 - N.base;
    const int C2 = pack_chunks(N.K);
    const int lane = (int)(w & 63), half = (int)((w >> 6) & 1);
    const long tc = w >> 7;
    const int c = (int)(tc % C2), t = (int)(tc / C2);
    const int m = 16 * t + (lane & 15), k = 32 * c + 8 * (lane >> 4) + 4 * half;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < N.M) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (k + r < N.K) v[r] = N.tr ? N.src[(size_t)(k + r) * N.ld + m] : N.src[(size_t)m * N.ld + k + r];
    }
    *(float4*)(P.packed + 4 * w0) = make_float4(v[0], v[1], v[2], v[3]);
}

// beta^n for Adam's bias corrections by repeated squaring: ~40 double multiplications instead of pow()'s several hundred
// instructions -- the two powers were 3 us of ONE lane with the whole launch waiting behind it (phase stamps: 7 k cycles
// at the barrier).  Within a few ulp of pow's double result; the coefficients are rounded to float afterwards.
__device__ __forceinline__ double ipow(double b, int n) {
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r *= b;
        b *= b;
        n >>= 1;
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// one optimiser group's step for the row schedule: Adam (smx_adam_step_dev_f32's expressions), the target update of the
// group's target network (smx_soft_update_f32 / smx_hard_update_every_f32's) and BOTH fragment-order copies, element by
// element in one launch -- the schedule's two pack launches and its target-update launch (12 + 4 us of a 135 us
// iteration) disappear.  Nothing reads the target critic between the critic's Adam step and the end of the iteration
// (the actor phase goes through the MODEL critic), so updating it here forms the same values as ddpg.py:344-352 does at
// the end.
// ---------------------------------------------------------------------------------------------------------------
struct UMat {
    long off;                     // first element of the matrix inside the group's buffer
    int M, K;                     // [M, K] row-major
    long base, base_tgt;          // packed blocks (float offsets): the model's copy, the target's
    long base_t0, base_t1;        // transposed copies (< 0: none): columns < split -> t0, the rest -> t1
    int split;
};
struct UArgs {
    float* theta;
    const float* grads;
    float* grads_out;             // the fused weight-gradient launch writes the gradient it steps with here
    float *m, *v, *target, *packed;
    long n;
    const float* lr;
    const int* step;
    float wd, clip_value, tau;
    int interval;
    UMat mat[3];
};

__global__ __launch_bounds__(256) void ddpg_rows_update_kernel(UArgs U) {
    __shared__ float coef[2];
    __shared__ int upd;
    if (threadIdx.x == 0) {
        const int st = *U.step;
        const double bc1 = 1.0 - ipow(0.9, st), bc2 = 1.0 - ipow(0.999, st);
        coef[0] = (float)(-((double)*U.lr / bc1));
        coef[1] = (float)sqrt(bc2);
        upd = U.interval > 0 ? (st % U.interval == 0) : 1;
    }
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= U.n) return;
    const float neg_step_size = coef[0], bc2_sqrt = coef[1];
    const float w1 = (float)(1.0 - 0.9), b2f = (float)0.999, w2 = (float)(1.0 - 0.999), eps = 1e-8f;
    float g = U.grads[i];
    if (U.clip_value > 0.f) g = fminf(fmaxf(g, -U.clip_value), U.clip_value);   // clip_grad_value_
    const float p = U.theta[i];
    if (U.wd != 0.f) g = g + U.wd * p;
    float mi = U.m[i], vi = U.v[i];
    mi = mi + w1 * (g - mi);
    vi = vi * b2f + w2 * (g * g);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pn = p + (neg_step_size * mi) / denom;
    U.theta[i] = pn;
    U.m[i] = mi;
    U.v[i] = vi;
    // the target network's element (torchx Module.soft_update: target * (1 - tau) + source * tau; a hard update copies)
    bool tw = false;
    float tn = 0.f;
    if (U.target && upd) {
        tn = (U.interval > 0 || U.tau >= 1.0f) ? pn : (U.target[i] * (1.0f - U.tau) + pn * U.tau);
        U.target[i] = tn;
        tw = true;
    }
    // the fragment-order copies
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const UMat& X = U.mat[j];
        const long r = i - X.off;
        if (r >= 0 && r < (long)X.M * X.K) {
            const int mm = (int)(r / X.K), kk = (int)(r - (long)mm * X.K);
            const long pos = pack_pos(X.K, mm, kk);
            U.packed[X.base + pos] = pn;
            if (tw) U.packed[X.base_tgt + pos] = tn;
            if (kk < X.split) {
                if (X.base_t0 >= 0) U.packed[X.base_t0 + pack_pos(X.M, kk, mm)] = pn;
            } else if (X.base_t1 >= 0) {
                U.packed[X.base_t1 + pack_pos(X.M, kk - X.split, mm)] = pn;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The group's weight gradients AND its step in one launch (single rank): dW = dz^T x and db = column sums of dz over the
// batch rows, then -- the gradient still in the accumulators -- the element's Adam step, target update and packed copies
// as in ddpg_rows_update_kernel (value clipping is elementwise: no norm over the group is needed first, which is what keeps
// PPO's clip-norm step a launch of its own).  Replaces smx_linear_multi_f32 (9 us at batch 512) + the update launch (6 us)
// and the gap between them.
//
// A workgroup owns a 32 x 32 tile of one matrix: wave w forms ALL of it over the w-th eighth of the rows on
// v_mfma_f32_16x16x4 (four quadrant accumulators; lane (i, g): a = dz[k + g][m0 + i], b = x[k + g][n0 + i], four rows
// per instruction), the operands of its 64 rows requested at once; the eight partial tiles meet through LDS in row order
// and every thread steps two elements.  Another summation order than smx_linear_wgrad_f32's: equal within fp32 rounding.
// ---------------------------------------------------------------------------------------------------------------
struct WMat {
    const float* dz;              // [rows][M], row stride ldz
    const float* x;               // [rows][N], row stride ldx
    int ldz, ldx, tiles_n, tile0; // tile0: first workgroup of this matrix
    long boff;                    // the bias [M] inside the group's buffer
};
struct WUArgs {
    UArgs U;
    WMat w[3];
    int rows;
    // optional: the iteration's statistics as one more workgroup of this launch (the last one of the iteration)
    float* stats;
    float* stats_host;            // host-mapped mirror, two slots of 8 floats: slot *step & 1
    const float *s_q, *s_y, *s_rewards, *s_actions, *s_q_actor;
    int s_A, tiles;
    long long* tbuf;              // SMX_DDPG_TIMING builds
};
#ifdef SMX_DDPG_TIMING
#define WSTAMP(i) do { if (G.tbuf && threadIdx.x == 0) G.tbuf[(size_t)blockIdx.x * 128 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif

#define MFMA16W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void ddpg_step_element(const UArgs& U, long i, float g, float p, float mi, float vi, float tg,
                                                  float neg_step_size, float bc2_sqrt, int upd, float& pn, float& tn,
                                                  bool& tw) {
    const float w1 = (float)(1.0 - 0.9), b2f = (float)0.999, w2 = (float)(1.0 - 0.999), eps = 1e-8f;
    if (U.clip_value > 0.f) g = fminf(fmaxf(g, -U.clip_value), U.clip_value);
    if (U.wd != 0.f) g = g + U.wd * p;
    mi = mi + w1 * (g - mi);
    vi = vi * b2f + w2 * (g * g);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pn = p + (neg_step_size * mi) / denom;
    U.theta[i] = pn;
    U.m[i] = mi;
    U.v[i] = vi;
    tw = false;
    tn = 0.f;
    if (U.target && upd) {
        tn = (U.interval > 0 || U.tau >= 1.0f) ? pn : (tg * (1.0f - U.tau) + pn * U.tau);
        U.target[i] = tn;
        tw = true;
    }
}

constexpr int WNW = 8;            // waves per tile: wave w takes the w-th eighth of the rows, the WHOLE 32 x 32 tile
constexpr int WUB = 16;           // steps (of four rows) a wave requests at once: 64 rows, 64 loads in flight

__global__ __launch_bounds__(64 * WNW) void ddpg_rows_wgrad_update_kernel(WUArgs G) {
    __shared__ float red[WNW][4][64][4];            // every wave's four quadrant accumulators (32 KB)
    __shared__ float redb[WNW][32];                 // ... and its column sums of dz
    __shared__ float coef[2];
    __shared__ int upd_s;
    const UArgs& U = G.U;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    if ((int)blockIdx.x == G.tiles) {               // the workgroup behind the last tile: the statistics
        float* mirror = G.stats_host ? G.stats_host + 8 * (*G.U.step & 1) : nullptr;
        ddpg_stats_block<64 * WNW>(G.s_q, G.s_y, G.s_rewards, G.s_actions, G.s_A, G.s_A, G.s_q_actor, (long)G.rows, G.stats,
                                   mirror);
        return;
    }
    WSTAMP(0);
    // consecutive workgroup ids land on the eight XCDs round-robin, each with its own L2: XCD x takes the x-th contiguous
    // eighth of the tile list (tiles are row-major in M: a band of dz columns and all of x) instead of every eighth tile,
    // or all eight L2s pull every operand across the fabric (as smx_gemm.hip's gemm32 does)
    int bid = blockIdx.x;
    {
        const int total = G.tiles, qd = total >> 3, rem = total & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + slot;
    }
    const int j = (bid >= G.w[2].tile0) ? 2 : (bid >= G.w[1].tile0) ? 1 : 0;
    const WMat W = G.w[j];
    const UMat X = U.mat[j];
    const int tile = bid - W.tile0;
    const int tm = tile / W.tiles_n, tn_ = tile - tm * W.tiles_n;
    const int m0 = 32 * tm, n0 = 32 * tn_;
    const int rows = G.rows;
    const int qrows = ((rows + 4 * WNW - 1) / (4 * WNW)) << 2;      // rows per wave (a multiple of four)
    const int k_lo = wv * qrows;
    int k_hi = k_lo + qrows;
    k_hi = k_hi < rows ? k_hi : rows;
    // addressing: the lane's part (row g of a step, its column) is a vector offset formed ONCE, the step's rows a SCALAR
    // offset -- no vector instruction between the loads.  The descriptors end at the wave's last row: the range check
    // counts the scalar offset in, so a row past k_hi (or a column past the matrix: OOB) loads 0.
    const rsrc_t rzk = make_rsrc(W.dz, (unsigned)(k_hi > 0 ? k_hi : 0) * (unsigned)W.ldz * 4u);
    const rsrc_t rxk = make_rsrc(W.x, (unsigned)(k_hi > 0 ? k_hi : 0) * (unsigned)W.ldx * 4u);
    const unsigned gz = (unsigned)g * (unsigned)W.ldz, gx = (unsigned)g * (unsigned)W.ldx;
    const unsigned vz0 = (m0 + i < X.M) ? (gz + (unsigned)(m0 + i)) * 4u : OOB;
    const unsigned vz1 = (m0 + 16 + i < X.M) ? (gz + (unsigned)(m0 + 16 + i)) * 4u : OOB;
    const unsigned vx0 = (n0 + i < X.K) ? (gx + (unsigned)(n0 + i)) * 4u : OOB;
    const unsigned vx1 = (n0 + 16 + i < X.K) ? (gx + (unsigned)(n0 + 16 + i)) * 4u : OOB;
    const unsigned sz = 16u * (unsigned)W.ldz, sx = 16u * (unsigned)W.ldx;      // bytes per step of four rows
#define SMX_LDW(R, v, k, sb) __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R, v, (unsigned)(k) / 4u * (sb), 0))
    // Every wave forms the whole tile over ITS rows: a step is four loads (two halves of the dz columns, two of the x
    // columns: each 64-byte segment is requested once per workgroup) for four products.  As quadrant waves over half the
    // rows each segment was requested twice and the launch waited on its 4096 requests per workgroup (phase stamps: the
    // last wave's operands arrived 17 k cycles in).
    float a0[WUB], a1[WUB], b0[WUB], b1[WUB];
#pragma unroll
    for (int u = 0; u < WUB; ++u) {
        a0[u] = SMX_LDW(rzk, vz0, k_lo + 4 * u, sz); a1[u] = SMX_LDW(rzk, vz1, k_lo + 4 * u, sz);
        b0[u] = SMX_LDW(rxk, vx0, k_lo + 4 * u, sx); b1[u] = SMX_LDW(rxk, vx1, k_lo + 4 * u, sx);
    }
    WSTAMP(1);
    // the two elements this THREAD steps once the waves' sums have met: (ml, nl) and (ml + 16, nl) of the tile, threads
    // along n (coalesced); their parameters and moments are requested behind the operands
    const int ml = tid >> 5, nl = tid & 31;
    long ei[2];
    float p2[2], m2[2], v2[2], t2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int m = m0 + ml + 16 * e, n = n0 + nl;
        const bool ok = m < X.M && n < X.K;
        ei[e] = ok ? X.off + (long)m * X.K + n : -1;
        const long c = ok ? ei[e] : X.off;
        p2[e] = U.theta[c]; m2[e] = U.m[c]; v2[e] = U.v[c];
        t2[e] = U.target ? U.target[c] : 0.f;
    }
    // the bias of row m0 + tid (column tile 0, the first 32 threads)
    const bool bok = tn_ == 0 && tid < 32 && m0 + tid < X.M;
    const long bi = bok ? W.boff + m0 + tid : W.boff;
    const float bp = U.theta[bi], bm = U.m[bi], bv = U.v[bi], bt = U.target ? U.target[bi] : 0.f;
    if (tid == 64 * (WNW - 1)) {                    // Adam's bias corrections, while the loads are on their way
        const int st = *U.step;
        const double bc1 = 1.0 - ipow(0.9, st), bc2 = 1.0 - ipow(0.999, st);
        coef[0] = (float)(-((double)*U.lr / bc1));
        coef[1] = (float)sqrt(bc2);
        upd_s = U.interval > 0 ? (st % U.interval == 0) : 1;
    }
    f32x4 acc[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) acc[qq] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float as0 = 0.f, as1 = 0.f;
    WSTAMP(2);
#pragma unroll 1
    for (int k = k_lo; k < k_hi; k += 4 * WUB) {
#pragma unroll
        for (int u = 0; u < WUB; ++u) {
            acc[0] = MFMA16W(a0[u], b0[u], acc[0]);
            acc[1] = MFMA16W(a0[u], b1[u], acc[1]);
            acc[2] = MFMA16W(a1[u], b0[u], acc[2]);
            acc[3] = MFMA16W(a1[u], b1[u], acc[3]);
            as0 += a0[u];
            as1 += a1[u];
        }
        if (k + 4 * WUB < k_hi) {                   // (more than 512 rows: the next 64 of this wave's)
#pragma unroll
            for (int u = 0; u < WUB; ++u) {
                a0[u] = SMX_LDW(rzk, vz0, k + 4 * (WUB + u), sz); a1[u] = SMX_LDW(rzk, vz1, k + 4 * (WUB + u), sz);
                b0[u] = SMX_LDW(rxk, vx0, k + 4 * (WUB + u), sx); b1[u] = SMX_LDW(rxk, vx1, k + 4 * (WUB + u), sx);
            }
        }
    }
#undef SMX_LDW
    WSTAMP(3);
    // quadrant qq = 2 (m half) + (n half): lane (i, g) holds rows 16 (qq >> 1) + 4 g + r, column 16 (qq & 1) + i
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) *(f32x4*)&red[wv][qq][lane][0] = acc[qq];
    as0 = meet_kq1(as0);                            // every lane of column i: the sum over this wave's rows
    as1 = meet_kq1(as1);
    if (g == 0) { redb[wv][i] = as0; redb[wv][16 + i] = as1; }
    __syncthreads();
    WSTAMP(4);
    const float neg_step_size = coef[0], bc2_sqrt = coef[1];
    const int upd = upd_s;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (ei[e] >= 0) {
            const int mm = ml + 16 * e;                                       // (inside the tile)
            const int qq = ((mm >> 4) << 1) | (nl >> 4), ln = 16 * ((mm & 15) >> 2) + (nl & 15), r = mm & 3;
            float gsum = red[0][qq][ln][r];
#pragma unroll
            for (int w = 1; w < WNW; ++w) gsum += red[w][qq][ln][r];          // the waves' row ranges in order
            U.grads_out[ei[e]] = gsum;
            float pn, tn;
            bool tw;
            ddpg_step_element(U, ei[e], gsum, p2[e], m2[e], v2[e], t2[e], neg_step_size, bc2_sqrt, upd, pn, tn, tw);
            const int gm = m0 + mm, kk = n0 + nl;
            const long pos = pack_pos(X.K, gm, kk);
            U.packed[X.base + pos] = pn;
            if (tw) U.packed[X.base_tgt + pos] = tn;
            if (kk < X.split) {
                if (X.base_t0 >= 0) U.packed[X.base_t0 + pack_pos(X.M, kk, gm)] = pn;
            } else if (X.base_t1 >= 0) {
                U.packed[X.base_t1 + pack_pos(X.M, kk - X.split, gm)] = pn;
            }
        }
    }
    if (bok) {
        float gb = redb[0][tid];
#pragma unroll
        for (int w = 1; w < WNW; ++w) gb += redb[w][tid];
        U.grads_out[bi] = gb;
        float pn, tn;
        bool tw;
        ddpg_step_element(U, bi, gb, bp, bm, bv, bt, neg_step_size, bc2_sqrt, upd, pn, tn, tw);
    }
    WSTAMP(5);
}

// block order of the packed buffer
enum { B_AW1, B_AW2, B_AW3, B_AW3T, B_AW2T, B_CW1, B_CW2, B_CW3, B_CW2TLO, B_CW2THI, B_TAW1, B_TAW2, B_TAW3, B_TCW1,
       B_TCW2, B_TCW3, B_COUNT };
static_assert(B_COUNT == N_MAT, "one item per block");

struct Dims { int D, A, H1, H2, c1, c2; };

void block_shape(const Dims& d, int b, int& M, int& K) {
    switch (b) {
        case B_AW1: case B_TAW1: M = d.H1; K = d.D; break;
        case B_AW2: case B_TAW2: M = d.H2; K = d.H1; break;
        case B_AW3: case B_TAW3: M = d.A; K = d.H2; break;
        case B_AW3T: M = d.H2; K = d.A; break;
        case B_AW2T: M = d.H1; K = d.H2; break;
        case B_CW1: case B_TCW1: M = d.c1; K = d.D; break;
        case B_CW2: case B_TCW2: M = d.c2; K = d.c1 + d.A; break;
        case B_CW3: case B_TCW3: M = 1; K = d.c2; break;
        case B_CW2TLO: M = d.c1; K = d.c2; break;
        default: M = d.A; K = d.c2; break;       // B_CW2THI
    }
}
long block_base(const Dims& d, int b) {          // in 16-byte words
    long o = 0;
    for (int k = 0; k < b; ++k) {
        int M, K;
        block_shape(d, k, M, K);
        o += pack_words(M, K);
    }
    return o;
}

long long* g_tbuf = nullptr;

int lds_floats(const Dims& d, RArgs* G) {
    constexpr int RB = RBLK;
    const int pad = 16;             // row strides = 16 mod 64: the 16 (row, kq) readers of the 4-row loop on distinct banks
    const int ldx = r64(d.D) + pad;
    const int wa = d.H1 > d.c2 ? d.H1 : d.c2, wb = d.H2 > d.c2 ? d.H2 : d.c2;
    const int wc = d.c1 + d.A > d.H1 ? d.c1 + d.A : d.H1;
    const int ldA = r64(wa) + pad, ldB = r64(wb) + pad, ldC = r64(wc) + pad;
    int o = 0;
    const int oX = o; o += RB * ldx;
    const int oXn = o; o += RB * ldx;
    const int oA = o; o += RB * ldA;
    const int oB = o; o += RB * ldB;
    const int oC = o; o += RB * ldC;
    const int oO = o; o += RB * LDO;
    const int oO2 = o; o += RB * LDO;
    const int oO3 = o; o += RB * LDO;
    const int oZ = o; o += RB * LDK4;
    const int oS = o; o += 16;
    const int oR = o; o += DNWV * 2 * 4 * 16;      // the split-K layers' partial sums
    o += 128;                     // the K loop's prefetch reads up to two chunks past a tile's last row
    if (G) {
        G->ldx = ldx; G->ldA = ldA; G->ldB = ldB; G->ldC = ldC;
        G->oX = oX; G->oXn = oXn; G->oA = oA; G->oB = oB; G->oC = oC; G->oO = oO; G->oO2 = oO2; G->oO3 = oO3;
        G->oZ = oZ; G->oS = oS; G->oR = oR; G->total = o;
    }
    return o;
}

bool dims_ok(const Dims& d) {
    return d.D > 0 && d.A > 0 && d.A <= 32 && d.H1 > 0 && d.H2 > 0 && d.c1 > 0 && d.c2 > 0 && d.D <= 2048 &&
           d.H1 % 4 == 0 && d.H2 % 4 == 0 && d.c1 % 4 == 0 && d.c2 % 4 == 0 && d.H1 <= 1024 && d.H2 <= 1024 &&
           d.c1 <= 1024 && d.c2 <= 1024 && lds_floats(d, nullptr) * (int)sizeof(float) <= MAX_LDS;
}

Dims dims_of(const smx_ddpg_rows_t& a) {
    Dims d;
    d.D = a.D; d.A = a.A; d.H1 = a.H1; d.H2 = a.H2; d.c1 = a.c1; d.c2 = a.c2;
    return d;
}

PMat pmat(const smx_ddpg_rows_t& a, const Dims& d, int b) {
    PMat m;
    block_shape(d, b, m.M, m.K);
    m.P = a.packed + 4 * block_base(d, b);
    return m;
}

int fill(RArgs& G, const smx_ddpg_rows_t* a) {
    SMX_REQUIRE(a && a->packed, SMX_E_NULL);
    const Dims d = dims_of(*a);
    SMX_REQUIRE(a->rows > 0 && a->rows < (1 << 24), SMX_E_SHAPE);
    SMX_REQUIRE(dims_ok(d), SMX_E_UNSUPPORTED);
    {   // every row-major output is addressed through a buffer descriptor: 31-bit byte offsets
        int widest = d.c1 + d.A;
        widest = d.H1 > widest ? d.H1 : widest;
        widest = d.c2 > widest ? d.c2 : widest;
        widest = d.H2 > widest ? d.H2 : widest;
        SMX_REQUIRE((int64_t)a->rows * widest * 4 < (1ll << 31), SMX_E_SHAPE);
    }
    SMX_REQUIRE(((uintptr_t)a->packed & 15) == 0, SMX_E_ALIGN);
    const smx_ddpg_net_t* nets[4] = {&a->actor, &a->critic, &a->target_actor, &a->target_critic};
    for (int k = 0; k < 4; ++k)
        SMX_REQUIRE(nets[k]->W1 && nets[k]->b1 && nets[k]->W2 && nets[k]->b2 && nets[k]->W3 && nets[k]->b3, SMX_E_NULL);
    memset(&G, 0, sizeof(G));
    G.rows = (int)a->rows; G.D = d.D; G.A = d.A; G.H1 = d.H1; G.H2 = d.H2; G.c1 = d.c1; G.c2 = d.c2;
    RNet* rn[4] = {&G.a, &G.c, &G.ta, &G.tc};
    const int w1[4] = {B_AW1, B_CW1, B_TAW1, B_TCW1};
    for (int k = 0; k < 4; ++k) {
        rn[k]->b1 = nets[k]->b1; rn[k]->b2 = nets[k]->b2; rn[k]->b3 = nets[k]->b3;
        rn[k]->W1 = pmat(*a, d, w1[k]);
        rn[k]->W2 = pmat(*a, d, w1[k] == B_AW1 ? B_AW2 : (w1[k] == B_CW1 ? B_CW2 : w1[k] + 1));
        rn[k]->W3 = pmat(*a, d, w1[k] == B_AW1 ? B_AW3 : (w1[k] == B_CW1 ? B_CW3 : w1[k] + 2));
    }
    G.cW2Tlo = pmat(*a, d, B_CW2TLO); G.cW2Thi = pmat(*a, d, B_CW2THI);
    G.aW3T = pmat(*a, d, B_AW3T); G.aW2T = pmat(*a, d, B_AW2T);
    G.cW3 = a->critic.W3;
    G.x = a->x; G.xn = a->x_next; G.actions = a->actions; G.rewards = a->rewards; G.dones = a->dones;
    G.gamma_n = a->gamma_n;
    G.xcat = a->xcat; G.h2c = a->h2c; G.q = a->q; G.q_next = a->q_next; G.y = a->y; G.dz3 = a->dz3; G.dz2 = a->dz2;
    G.dxcat = a->dxcat; G.h1a = a->h1a; G.h2a = a->h2a; G.act = a->act;
    G.q_actor = a->q_actor; G.dz3a = a->dz3a; G.dz2a = a->dz2a; G.dz1a = a->dz1a;
    G.step = a->step;
    lds_floats(d, &G);
    G.tbuf = g_tbuf;
    return SMX_OK;
}

Step step(int in_off, int ldi, const PMat& W, const float* bias, int act, int out_off, int ldo, float* g, int ldg, int post) {
    Step S;
    memset(&S, 0, sizeof(S));
    S.W = W.P; S.bias = bias; S.g = g; S.M = W.M; S.K = W.K; S.in_off = in_off; S.ldi = ldi; S.act = act;
    S.out_off = out_off; S.ldo = ldo; S.ldg = ldg; S.post = post;
    return S;
}

// the two chains, step for step
void critic_program(const RArgs& G, Prog& P) {
    const int ldc = G.c1 + G.A;
    int n = 0;
    P.s[n++] = step(G.oXn, G.ldx, G.ta.W1, G.ta.b1, A_RELU, G.oA, G.ldA, nullptr, 0, P_NONE);         // mu'(s')
    P.s[n++] = step(G.oA, G.ldA, G.ta.W2, G.ta.b2, A_RELU, G.oB, G.ldB, nullptr, 0, P_NONE);
    P.s[n++] = step(G.oB, G.ldB, G.ta.W3, G.ta.b3, A_TANH, G.oO, LDO, nullptr, 0, P_NONE);
    P.s[n++] = step(G.oXn, G.ldx, G.tc.W1, G.tc.b1, A_RELU, G.oC, G.ldC, nullptr, 0, P_TC_CAT);       // Q'(s', mu'(s'))
    P.s[n++] = step(G.oC, G.ldC, G.tc.W2, G.tc.b2, A_RELU, G.oB, G.ldB, nullptr, 0, P_NONE);
    P.s[n++] = step(G.oB, G.ldB, G.tc.W3, G.tc.b3, A_NONE, G.oO2, LDO, nullptr, 0, P_NONE);
    P.s[n++] = step(G.oX, G.ldx, G.c.W1, G.c.b1, A_RELU, G.oC, G.ldC, G.xcat, ldc, P_C_CAT);          // Q(s, a)
    P.s[n++] = step(G.oC, G.ldC, G.c.W2, G.c.b2, A_RELU, G.oB, G.ldB, G.h2c, G.c2, P_NONE);
    P.s[n++] = step(G.oB, G.ldB, G.c.W3, G.c.b3, A_NONE, G.oO, LDO, nullptr, 0, P_LOSS);              // -> y, dz3, dz2
    P.s[n] = step(G.oA, G.ldA, G.cW2Tlo, nullptr, A_MASK, -1, 0, G.dxcat, ldc, P_NONE);               // dz1
    P.s[n].mask_off = G.oC; P.s[n].ldm = G.ldC; ++n;
    P.s[n++] = step(G.oX, G.ldx, G.a.W1, G.a.b1, A_RELU, G.oA, G.ldA, G.h1a, G.H1, P_NONE);           // mu(s), kept
    P.s[n++] = step(G.oA, G.ldA, G.a.W2, G.a.b2, A_RELU, G.oB, G.ldB, G.h2a, G.H2, P_NONE);
    P.s[n++] = step(G.oB, G.ldB, G.a.W3, G.a.b3, A_TANH, -1, 0, G.act, G.A, P_NONE);
    P.n = n;
}

void actor_program(const RArgs& G, Prog& P) {
    int n = 0;
    P.s[n++] = step(G.oX, G.ldx, G.c.W1, G.c.b1, A_RELU, G.oC, G.ldC, nullptr, 0, P_A_CAT);           // Q(s, mu(s))
    P.s[n++] = step(G.oC, G.ldC, G.c.W2, G.c.b2, A_RELU, G.oB, G.ldB, nullptr, 0, P_NONE);
    P.s[n++] = step(G.oB, G.ldB, G.c.W3, G.c.b3, A_NONE, -1, 0, G.q_actor, 1, P_A_DQ);                // -> masks, dz2
    P.s[n++] = step(G.oA, G.ldA, G.cW2Thi, nullptr, A_NONE, G.oO2, LDO, nullptr, 0, P_A_TANH);        // d/d(action)
    P.s[n] = step(G.oZ, LDK4, G.aW3T, nullptr, A_MASK, G.oA, G.ldA, G.dz2a, G.H2, P_NONE);            // the actor's dz2, dz1
    P.s[n].mask_off = G.oB; P.s[n].ldm = G.ldB; ++n;
    P.s[n] = step(G.oA, G.ldA, G.aW2T, nullptr, A_MASK, -1, 0, G.dz1a, G.H1, P_NONE);
    P.s[n].mask_off = G.oC; P.s[n].ldm = G.ldC; ++n;
    P.n = n;
}
static_assert(MAX_STEPS >= 13, "the critic chain has 13 layers");

int set_lds(const void* fn, int bytes) {
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

extern "C" void smx_ddpg_rows_debug_tbuf(void* p) { g_tbuf = (long long*)p; }

extern "C" int32_t smx_ddpg_rows_supported(int32_t D, int32_t A, int32_t H1, int32_t H2, int32_t c1, int32_t c2) {
    Dims d;
    d.D = D; d.A = A; d.H1 = H1; d.H2 = H2; d.c1 = c1; d.c2 = c2;
    return dims_ok(d) ? 1 : 0;
}

extern "C" int32_t smx_ddpg_rows_supported_at(int32_t D, int32_t A, int32_t H1, int32_t H2, int32_t c1, int32_t c2,
                                              int64_t rows) {
    Dims d;
    d.D = D; d.A = A; d.H1 = H1; d.H2 = H2; d.c1 = c1; d.c2 = c2;
    return rows > 0 && rows < (1 << 24) && dims_ok(d) ? 1 : 0;
}

extern "C" int64_t smx_ddpg_rows_packed_floats(int32_t D, int32_t A, int32_t H1, int32_t H2, int32_t c1, int32_t c2) {
    Dims d;
    d.D = D; d.A = A; d.H1 = H1; d.H2 = H2; d.c1 = c1; d.c2 = c2;
    return dims_ok(d) ? 4 * block_base(d, B_COUNT) : 0;
}

extern "C" int smx_ddpg_rows_pack_f32(const smx_ddpg_rows_t* a, int32_t which, smx_stream_t stream) {
    SMX_REQUIRE(a && a->packed, SMX_E_NULL);
    const Dims d = dims_of(*a);
    SMX_REQUIRE(dims_ok(d), SMX_E_UNSUPPORTED);
    SMX_REQUIRE(which == SMX_DDPG_PACK_ALL || which == SMX_DDPG_PACK_CRITIC, SMX_E_SHAPE);
    const int ldc = d.c1 + d.A;
    const float* src[B_COUNT] = {a->actor.W1, a->actor.W2, a->actor.W3, a->actor.W3, a->actor.W2,
                                 a->critic.W1, a->critic.W2, a->critic.W3, a->critic.W2, a->critic.W2 + d.c1,
                                 a->target_actor.W1, a->target_actor.W2, a->target_actor.W3,
                                 a->target_critic.W1, a->target_critic.W2, a->target_critic.W3};
    const int ld[B_COUNT] = {d.D, d.H1, d.H2, d.H2, d.H1, d.D, ldc, d.c2, ldc, ldc, d.D, d.H1, d.H2, d.D, ldc, d.c2};
    const int tr[B_COUNT] = {0, 0, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0};
    const int b0 = which == SMX_DDPG_PACK_CRITIC ? B_CW1 : 0, b1 = which == SMX_DDPG_PACK_CRITIC ? B_CW2THI + 1 : B_COUNT;
    PArgs P;
    memset(&P, 0, sizeof(P));
    P.packed = a->packed;
    P.count = b1 - b0;
    for (int b = b0; b < b1; ++b) {
        SMX_REQUIRE(src[b], SMX_E_NULL);
        PItem& it = P.it[b - b0];
        it.src = src[b]; it.ld = ld[b]; it.tr = tr[b];
        block_shape(d, b, it.M, it.K);
        it.base = block_base(d, b);
    }
    P.total = block_base(d, b1) - block_base(d, b0);
    hipLaunchKernelGGL(ddpg_pack_kernel, dim3((unsigned)((P.total + 255) / 256)), dim3(256), 0, smx_s(stream), P);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ddpg_rows_critic_f32(const smx_ddpg_rows_t* a, smx_stream_t stream) {
    RArgs G;
    const int rc = fill(G, a);
    if (rc) return rc;
    SMX_REQUIRE(a->x && a->x_next && a->actions && a->rewards && a->dones, SMX_E_NULL);
    SMX_REQUIRE(a->xcat && a->h2c && a->q && a->q_next && a->y && a->dz3 && a->dz2 && a->dxcat && a->h1a && a->h2a &&
                    a->act, SMX_E_NULL);
    const int bytes = G.total * (int)sizeof(float);
    static int set = 0;
    if (set < bytes) {
        const int e = set_lds((const void*)ddpg_rows4_kernel<0>, bytes);
        if (e) return e;
        set = bytes;
    }
    Prog P;
    memset(&P, 0, sizeof(P));
    critic_program(G, P);
    hipLaunchKernelGGL(ddpg_rows4_kernel<0>, dim3((unsigned)((G.rows + RBLK - 1) / RBLK)), dim3(DNTH), bytes, smx_s(stream), G, P);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ddpg_rows_actor_f32(const smx_ddpg_rows_t* a, smx_stream_t stream) {
    RArgs G;
    const int rc = fill(G, a);
    if (rc) return rc;
    SMX_REQUIRE(a->x && a->h1a && a->h2a && a->act && a->q_actor && a->dz3a && a->dz2a && a->dz1a, SMX_E_NULL);
    const int bytes = G.total * (int)sizeof(float);
    static int set = 0;
    if (set < bytes) {
        const int e = set_lds((const void*)ddpg_rows4_kernel<1>, bytes);
        if (e) return e;
        set = bytes;
    }
    Prog P;
    memset(&P, 0, sizeof(P));
    actor_program(G, P);
    hipLaunchKernelGGL(ddpg_rows4_kernel<1>, dim3((unsigned)((G.rows + RBLK - 1) / RBLK)), dim3(DNTH), bytes, smx_s(stream), G, P);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

namespace {
int fill_update(UArgs& U, const smx_ddpg_rows_t* a, int32_t group, const smx_ddpg_update_t* u, const Dims& d) {
    SMX_REQUIRE(u->theta && u->grads && u->exp_avg && u->exp_avg_sq && u->lr && u->step, SMX_E_NULL);
    SMX_REQUIRE(group == SMX_DDPG_GROUP_ACTOR || group == SMX_DDPG_GROUP_CRITIC, SMX_E_SHAPE);
    SMX_REQUIRE(u->n > 0 && u->interval >= 0 && (u->target == nullptr || u->interval > 0 || u->tau > 0.f), SMX_E_SHAPE);
    const bool cr = group == SMX_DDPG_GROUP_CRITIC;
    const smx_ddpg_net_t& net = cr ? a->critic : a->actor;
    const smx_ddpg_net_t& tnet = cr ? a->target_critic : a->target_actor;
    SMX_REQUIRE(net.W1 && net.W2 && net.W3, SMX_E_NULL);
    memset(&U, 0, sizeof(U));
    U.theta = u->theta; U.grads = u->grads; U.m = u->exp_avg; U.v = u->exp_avg_sq; U.target = u->target;
    U.packed = a->packed; U.n = u->n; U.lr = u->lr; U.step = u->step; U.wd = u->weight_decay;
    U.clip_value = u->clip_value; U.tau = u->tau; U.interval = u->interval;
    const float* W[3] = {net.W1, net.W2, net.W3};
    const float* TW[3] = {tnet.W1, tnet.W2, tnet.W3};
    const int blk[3] = {cr ? B_CW1 : B_AW1, cr ? B_CW2 : B_AW2, cr ? B_CW3 : B_AW3};
    const int tblk[3] = {cr ? B_TCW1 : B_TAW1, cr ? B_TCW2 : B_TAW2, cr ? B_TCW3 : B_TAW3};
    for (int j = 0; j < 3; ++j) {
        UMat& X = U.mat[j];
        block_shape(d, blk[j], X.M, X.K);
        X.off = W[j] - u->theta;
        // the matrices lie inside the group's buffer, the target's at the same offsets inside the target's
        SMX_REQUIRE(X.off >= 0 && X.off + (long)X.M * X.K <= u->n, SMX_E_SHAPE);
        if (u->target) SMX_REQUIRE(TW[j] && TW[j] - u->target == X.off, SMX_E_SHAPE);
        X.base = 4 * block_base(d, blk[j]);
        X.base_tgt = 4 * block_base(d, tblk[j]);
        X.base_t0 = X.base_t1 = -1;
        X.split = X.K;
    }
    if (cr) {
        U.mat[1].base_t0 = 4 * block_base(d, B_CW2TLO);
        U.mat[1].base_t1 = 4 * block_base(d, B_CW2THI);
        U.mat[1].split = d.c1;
    } else {
        U.mat[1].base_t0 = 4 * block_base(d, B_AW2T);
        U.mat[2].base_t0 = 4 * block_base(d, B_AW3T);
    }
    return SMX_OK;
}
}  // namespace

extern "C" int smx_ddpg_rows_update_f32(const smx_ddpg_rows_t* a, int32_t group, const smx_ddpg_update_t* u,
                                        smx_stream_t stream) {
    SMX_REQUIRE(a && a->packed && u, SMX_E_NULL);
    const Dims d = dims_of(*a);
    SMX_REQUIRE(dims_ok(d), SMX_E_UNSUPPORTED);
    UArgs U;
    const int rc = fill_update(U, a, group, u, d);
    if (rc) return rc;
    hipLaunchKernelGGL(ddpg_rows_update_kernel, dim3((unsigned)((u->n + 255) / 256)), dim3(256), 0, smx_s(stream), U);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_ddpg_rows_wgrad_update_f32(const smx_ddpg_rows_t* a, int32_t group, const smx_ddpg_update_t* u,
                                              smx_stream_t stream) {
    SMX_REQUIRE(a && a->packed && u, SMX_E_NULL);
    const Dims d = dims_of(*a);
    SMX_REQUIRE(dims_ok(d), SMX_E_UNSUPPORTED);
    SMX_REQUIRE(a->rows > 0 && a->rows < (1 << 24), SMX_E_SHAPE);
    WUArgs G;
    memset(&G, 0, sizeof(G));
    const int rc = fill_update(G.U, a, group, u, d);
    if (rc) return rc;
    G.U.grads_out = const_cast<float*>(u->grads);
    G.rows = (int)a->rows;
    G.tbuf = g_tbuf;
    const bool cr = group == SMX_DDPG_GROUP_CRITIC;
    const smx_ddpg_net_t& net = cr ? a->critic : a->actor;
    const smx_ddpg_net_t& tnet = cr ? a->target_critic : a->target_actor;
    const int ldc = d.c1 + d.A;
    // (gradient, input) of the three layers: the buffers the chain launches wrote
    const float* dz[3] = {cr ? a->dxcat : a->dz1a, cr ? a->dz2 : a->dz2a, cr ? a->dz3 : a->dz3a};
    const int ldz[3] = {cr ? ldc : d.H1, cr ? d.c2 : d.H2, cr ? 1 : d.A};
    const float* x[3] = {a->x, cr ? a->xcat : a->h1a, cr ? a->h2c : a->h2a};
    const int ldx[3] = {d.D, cr ? ldc : d.H1, cr ? d.c2 : d.H2};
    const float* b[3] = {net.b1, net.b2, net.b3};
    const float* tb[3] = {tnet.b1, tnet.b2, tnet.b3};
    int tiles = 0;
    for (int j = 0; j < 3; ++j) {
        SMX_REQUIRE(dz[j] && x[j] && b[j], SMX_E_NULL);
        WMat& W = G.w[j];
        const UMat& X = G.U.mat[j];
        W.dz = dz[j]; W.ldz = ldz[j]; W.x = x[j]; W.ldx = ldx[j];
        W.boff = b[j] - u->theta;
        SMX_REQUIRE(W.boff >= 0 && W.boff + X.M <= u->n, SMX_E_SHAPE);
        if (u->target) SMX_REQUIRE(tb[j] && tb[j] - u->target == W.boff, SMX_E_SHAPE);
        SMX_REQUIRE((int64_t)a->rows * (ldz[j] > ldx[j] ? ldz[j] : ldx[j]) * 4 < (1ll << 31), SMX_E_SHAPE);
        W.tiles_n = (X.K + 31) / 32;
        W.tile0 = tiles;
        tiles += ((X.M + 31) / 32) * W.tiles_n;
    }
    // every element of the group's buffer is a weight or a bias of the three layers: nothing is left without its step
    {
        long covered = 0;
        for (int j = 0; j < 3; ++j) covered += (long)G.U.mat[j].M * G.U.mat[j].K + G.U.mat[j].M;
        SMX_REQUIRE(covered == u->n, SMX_E_SHAPE);
    }
    G.tiles = tiles;
    if (u->stats) {
        SMX_REQUIRE(a->q && a->y && a->rewards && a->actions && a->q_actor, SMX_E_NULL);
        G.stats = u->stats; G.s_q = a->q; G.s_y = a->y; G.s_rewards = a->rewards; G.s_actions = a->actions;
        G.s_q_actor = a->q_actor; G.s_A = d.A;
        G.stats_host = u->stats_host;
    }
    hipLaunchKernelGGL(ddpg_rows_wgrad_update_kernel, dim3((unsigned)(tiles + (u->stats ? 1 : 0))), dim3(64 * WNW), 0,
                       smx_s(stream), G);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
