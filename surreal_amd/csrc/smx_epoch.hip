// Fused row-block kernels for the small-batch PPO epoch loops
// (surreal/learner/ppo.py:194-353 _clip/_adapt/_value loss + update, 541-562 the epoch loops;
// surreal/model/ppo_net.py:253-315 forward_actor / forward_critic; builders.py:86-175).
//
// An epoch runs the actor and the critic on B = 1024 rows: ~2 GFLOP, far too little to tile for
// reuse, and with one launch per layer the step was a chain of ~9 dependent launches per epoch,
// each re-reading its operands from L2 as 32x32 tiles (61 MB for a 0.46 GFLOP layer).  Here a
// workgroup owns 16 data rows of ONE network and carries them through the whole chain:
//
//   epoch_fwd_kernel   x tile -> LDS; layer 1 -> layer 2 -> layer 3 on v_mfma_f32_16x16x4_f32 in
//                      TRANSPOSED form (h^T = W . x^T: the MFMA M axis is the output feature, the
//                      N axis the data row), the four waves split the feature tiles, weights go
//                      L2 -> registers in fragment order straight from their row-major home (no
//                      packing: a lane reads 32 contiguous bytes of its weight row per 32-wide K
//                      chunk, a full 128-byte line per row across the wave), activations go
//                      wave -> LDS -> all waves between layers and to HBM once, transposed
//                      ([features, rows]: what the weight-gradient GEMMs read K-contiguously);
//                      then the loss of the job on the rows it holds: DiagGauss likelihoods / KL /
//                      surrogate (+ per-row gradient tiles and block partial sums) for the actor,
//                      squared error (+ dz3 and mergeable moments) for the critic.
//   epoch_bwd_kernel   batch means from the block partials -> KL coefficient, statistics, KL early
//                      exit (every workgroup reduces the same partial rows in the same order);
//                      dz3 -> dz2 = (W3^T dz3) * relu'(h2) -> dz1 = (W2^T dz2) * relu'(h1), again
//                      16 rows per workgroup, two workgroups per row block splitting the dz1
//                      feature tiles (the dz2 product is cheap and recomputed by both).
//
// The weight gradients (sums over ALL rows) stay a GEMM launch (smx_gemm.hip), clip-norm + Adam
// one more: 4 dependent launches per epoch instead of 9.
#include "smx_common.h"
#include <string.h>

namespace {
#include "smx_ppo_loss.inc.h"

constexpr int ER = 16;            // data rows per workgroup (= MFMA N)
constexpr int NWV = 4;            // waves per workgroup, one per SIMD
constexpr int NTH = 64 * NWV;
constexpr int TG = 5;             // feature tiles a wave carries per pass (4 accumulator VGPRs each)
constexpr int MAX_EJOBS = 4;
constexpr int LDO = 36;           // row stride of the output tile in LDS (<= 32 outputs)
static_assert(ER == LOSS_ROWS_PER_BLOCK, "a row block is a loss block");
static_assert(NTH == 256, "the shared loss code strides by 256 threads");

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;   // past every extent: the load returns 0, no traffic

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    const uintptr_t u = (uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void* q = (void*)(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__device__ __forceinline__ float4 ld16(rsrc_t R, unsigned off) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(R, off, 0, 0);
    float4 v;
    v.x = __uint_as_float(w.x); v.y = __uint_as_float(w.y);
    v.z = __uint_as_float(w.z); v.w = __uint_as_float(w.w);
    return v;
}
__device__ __forceinline__ float ld4(rsrc_t R, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R, off, 0, 0));
}

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == SMX_ACT_RELU) return (v < 0.f) ? 0.f : v;
    if (act == SMX_ACT_TANH) return tanhf(v);
    return v;
}

struct EJob {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int D, H1, H2, OUT;
    const float* x;
    int rows;
    float *h1T, *h2T;
    long ldT;
    float* out;
    int out_ld, out_act, loss, blk_base;
    const int* stop;
    // backward
    const float* dz3;
    float *dz3T, *dz2T, *dz1T;
};

struct PolArgs {       // the DiagGauss losses of the actor job (smx_ppo_losses_t, the parts used here)
    int mode, A;
    const float *log_var, *actions, *behave, *ref, *adv;
    int ld_act, ld_beh, ld_ref, check_stop, will_update;
    float *g_surr, *g_kl, *row_partials;
    float *dlogvar, *dlogvar_sumsq, *stats;
};

struct ValArgs {
    const float* returns;
    float *v_dz3, *v_partials;
    int will_update;
};

struct EArgs {
    EJob j[MAX_EJOBS];
    int n;
    PolArgs pl;
    ValArgs vl;
    long n_total;
    // LDS carve-up (floats), the same for every workgroup of the launch
    int ldx, ldh1, ldh2, off_h1, off_h2, off_out, off_red, off_loss;
    int fsplit;
};

__device__ __forceinline__ EJob select_job(const EArgs& G, int bid) {
    int pi = 0;
#pragma unroll
    for (int k = 1; k < MAX_EJOBS; ++k) pi += (k < G.n && bid >= G.j[k].blk_base) ? 1 : 0;
    EJob J = G.j[pi];
    // the whole descriptor in one batch of scalar loads (fetched field by field where first used
    // they form a chain of dependent kernarg round trips in front of the first operand load)
    asm volatile("" :: "s"(J.W1), "s"(J.b1), "s"(J.W2), "s"(J.b2), "s"(J.W3), "s"(J.b3), "s"(J.D), "s"(J.H1),
                 "s"(J.H2), "s"(J.OUT), "s"(J.x), "s"(J.rows), "s"(J.h1T), "s"(J.h2T), "s"(J.ldT), "s"(J.out),
                 "s"(J.out_ld), "s"(J.out_act), "s"(J.loss), "s"(J.blk_base), "s"(J.stop), "s"(J.dz3),
                 "s"(J.dz3T), "s"(J.dz2T), "s"(J.dz1T));
    return J;
}

// ---------------------------------------------------------------------------------------------
// forward: accT[g] (16 features x 16 rows) += W[16 t_g .. +16, :K] . inT[:K, 16 rows]
// lane l: fm = l & 15 is the feature within the tile for the A operand and the data row for the
// B operand and the C fragment; kq = l >> 4.  A 32-wide K chunk c is eight MFMA steps (h, r),
// h = 0/1, r = 0..3, step (h, r) multiplies k = 32c + 8kq + 4h + r: each lane fetches 32 contiguous
// bytes of its weight row and two 16-byte words of its data row per chunk.  Weight rows >= M and
// bytes past the matrix are out-of-range buffer loads (0); bytes past K inside the matrix belong to
// the next row and meet the zero padding of the data tile.
// ---------------------------------------------------------------------------------------------
template <int NT>
struct WFrag {
    float4 a[NT], b[NT];      // k = 8kq + 0..3 and 8kq + 4..7 of the chunk, per tile
};

template <int NT>
__device__ __forceinline__ void ld_wfrag(WFrag<NT>& f, rsrc_t rw, const unsigned (&wo)[TG], int c, int nch) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const unsigned o = (c < nch) ? wo[g] + (unsigned)c * 128u : OOB;
        f.a[g] = ld16(rw, o);
        f.b[g] = ld16(rw, o + 16u);
    }
}

template <int NT>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[TG], const WFrag<NT>& f, const float* bp) {
    const float4 b0 = *(const float4*)(bp);
    const float4 b1 = *(const float4*)(bp + 4);
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.a[g].x, b0.x, acc[g]);
        acc[g] = MFMA16(f.a[g].y, b0.y, acc[g]);
        acc[g] = MFMA16(f.a[g].z, b0.z, acc[g]);
        acc[g] = MFMA16(f.a[g].w, b0.w, acc[g]);
    }
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.b[g].x, b1.x, acc[g]);
        acc[g] = MFMA16(f.b[g].y, b1.y, acc[g]);
        acc[g] = MFMA16(f.b[g].z, b1.z, acc[g]);
        acc[g] = MFMA16(f.b[g].w, b1.w, acc[g]);
    }
}

// tiles t0, t0 + tstep, ... (NT of them) over the K chunks c0, c0 + cstep, ... < nch
template <int NT>
__device__ __forceinline__ void fwd_tiles(f32x4 (&acc)[TG], rsrc_t rw, int M, int K, const float* in_lds,
                                          int ldi, int t0, int tstep, int c0, int cstep, int fm, int kq) {
    const int nch = (K + 31) >> 5;
    unsigned wo[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g) {
        const int row = 16 * (t0 + tstep * g) + fm;
        wo[g] = (g < NT && row < M) ? ((unsigned)row * (unsigned)K + 8u * kq) * 4u : OOB;
    }
    const float* bp = in_lds + fm * ldi + 8 * kq;
    WFrag<NT> P, Q;
    ld_wfrag<NT>(P, rw, wo, c0, nch);
    for (int c = c0; c < nch; c += 2 * cstep) {
        ld_wfrag<NT>(Q, rw, wo, c + cstep, nch);
        mma_chunk<NT>(acc, P, bp + 32 * c);
        ld_wfrag<NT>(P, rw, wo, c + 2 * cstep, nch);
        if (c + cstep < nch) mma_chunk<NT>(acc, Q, bp + 32 * (c + cstep));
    }
}

__device__ __forceinline__ void fwd_tiles_n(int nt, f32x4 (&acc)[TG], rsrc_t rw, int M, int K,
                                            const float* in_lds, int ldi, int t0, int tstep, int c0,
                                            int cstep, int fm, int kq) {
    switch (nt) {     // wave-uniform
        case 1: fwd_tiles<1>(acc, rw, M, K, in_lds, ldi, t0, tstep, c0, cstep, fm, kq); break;
        case 2: fwd_tiles<2>(acc, rw, M, K, in_lds, ldi, t0, tstep, c0, cstep, fm, kq); break;
        case 3: fwd_tiles<3>(acc, rw, M, K, in_lds, ldi, t0, tstep, c0, cstep, fm, kq); break;
        case 4: fwd_tiles<4>(acc, rw, M, K, in_lds, ldi, t0, tstep, c0, cstep, fm, kq); break;
        case 5: fwd_tiles<5>(acc, rw, M, K, in_lds, ldi, t0, tstep, c0, cstep, fm, kq); break;
        default: break;
    }
}

// one hidden layer: every wave takes the feature tiles wv, wv + NWV, ...; bias + ReLU in the
// fragment; the tile goes to LDS as [row][feature] (the next layer's B operand) and to HBM as
// [feature][row] (the weight-gradient GEMM's K-contiguous operand; also the ReLU mask of the
// backward kernel)
__device__ __forceinline__ void hidden_layer(const float* W, const float* bias, int H, int K,
                                             const float* in_lds, int ldi, float* out_lds, int ldo,
                                             float* hT, long ldT, long row0, int nrows, int wv, int fm,
                                             int kq) {
    const int tiles = (H + 15) >> 4;
    const rsrc_t rw = make_rsrc(W, (unsigned)H * (unsigned)K * 4u);
    for (int tb = 0; tb < tiles; tb += NWV * TG) {
        const int t0 = tb + wv;
        int nt = (tiles - t0 + NWV - 1) / NWV;
        nt = nt < 0 ? 0 : (nt > TG ? TG : nt);
        // the epilogue's bias words are requested in front of the main loop
        float4 bs[TG];
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            const int f0 = 16 * (t0 + NWV * g) + 4 * kq;
            bs[g] = (g < nt && f0 < H) ? *(const float4*)(bias + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x4 acc[TG];
#pragma unroll
        for (int g = 0; g < TG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fwd_tiles_n(nt, acc, rw, H, K, in_lds, ldi, t0, NWV, 0, 1, fm, kq);
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            if (g < nt) {
                const int f0 = 16 * (t0 + NWV * g) + 4 * kq;     // features f0..f0+3 of data row fm
                float4 v;
                v.x = act_f(acc[g][0] + bs[g].x, SMX_ACT_RELU);
                v.y = act_f(acc[g][1] + bs[g].y, SMX_ACT_RELU);
                v.z = act_f(acc[g][2] + bs[g].z, SMX_ACT_RELU);
                v.w = act_f(acc[g][3] + bs[g].w, SMX_ACT_RELU);
                if (f0 >= H) v = make_float4(0.f, 0.f, 0.f, 0.f);      // H % 4 == 0: all four or none
                *(float4*)(out_lds + fm * ldo + f0) = v;
                if (hT && f0 < H && fm < nrows) {
                    float* q = hT + (size_t)f0 * ldT + row0 + fm;
                    q[0] = v.x; q[ldT] = v.y; q[2 * ldT] = v.z; q[3 * ldT] = v.w;
                }
            }
        }
    }
}

__global__ __launch_bounds__(NTH) void epoch_fwd_kernel(EArgs G, smx_ppo_ctrl_t* __restrict__ ctrl) {
    extern __shared__ float sm[];
    const EJob J = select_job(G, (int)blockIdx.x);
    const int stopv = J.stop ? __builtin_nontemporal_load(J.stop) : 0;
    const int blk = blockIdx.x - J.blk_base;
    const long row0 = (long)blk * ER;
    int nrows = J.rows - (int)row0;
    if (nrows > ER) nrows = ER;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 15, kq = lane >> 4;

    float* xs = sm;
    float* h1s = sm + G.off_h1;
    float* h2s = sm + G.off_h2;
    float* outs = sm + G.off_out;
    float* red = sm + G.off_red;
    const int ldx = G.ldx, ldh1 = G.ldh1, ldh2 = G.ldh2;

    // ---- x tile -> LDS (rows past the batch and k >= D are zero), hidden tiles cleared -------
    {
        const int D4 = J.D >> 2, X4 = ldx >> 2;
        for (int idx = tid; idx < ER * X4; idx += NTH) {
            const int n = idx / X4, j = idx - n * X4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < D4 && n < nrows) v = *(const float4*)(J.x + (size_t)(row0 + n) * J.D + 4 * j);
            *(float4*)(xs + n * ldx + 4 * j) = v;
        }
        for (int idx = tid; idx < (G.off_red - G.off_h1) >> 2; idx += NTH)
            *(float4*)(h1s + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (__builtin_amdgcn_readfirstlane(stopv) != 0) return;
    __syncthreads();

    hidden_layer(J.W1, J.b1, J.H1, J.D, xs, ldx, h1s, ldh1, J.h1T, J.ldT, row0, nrows, wv, fm, kq);
    __syncthreads();
    hidden_layer(J.W2, J.b2, J.H2, J.H1, h1s, ldh1, h2s, ldh2, J.h2T, J.ldT, row0, nrows, wv, fm, kq);
    __syncthreads();

    // ---- output layer: <= 2 feature tiles, the four waves split K, partial tiles meet in LDS ----
    {
        const int tiles = (J.OUT + 15) >> 4;
        const rsrc_t rw = make_rsrc(J.W3, (unsigned)J.OUT * (unsigned)J.H2 * 4u);
        f32x4 acc[TG];
#pragma unroll
        for (int g = 0; g < TG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fwd_tiles_n(tiles, acc, rw, J.OUT, J.H2, h2s, ldh2, 0, 1, wv, NWV, fm, kq);
#pragma unroll
        for (int g = 0; g < 2; ++g)
            if (g < tiles) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((wv * 2 + g) * 16 + 4 * kq + r) * 16 + fm] = acc[g][r];
            }
        __syncthreads();
        const int OP = 16 * tiles;
        for (int idx = tid; idx < ER * OP; idx += NTH) {
            const int n = idx / OP, f = idx - n * OP;
            const int e = ((f >> 4) * 16 + (f & 15)) * 16 + n;
            float v = ((red[e] + red[512 + e]) + red[1024 + e]) + red[1536 + e];
            if (f < J.OUT) {
                v = act_f(v + J.b3[f], J.out_act);
                if (J.out && n < nrows) J.out[(size_t)(row0 + n) * J.out_ld + f] = v;
            } else {
                v = 0.f;
            }
            outs[n * LDO + f] = v;
        }
        __syncthreads();
    }

    // ---- the job's loss on the rows it holds -------------------------------------------------
    if (J.loss == SMX_EPOCH_LOSS_POLICY) {
        const PolArgs& p = G.pl;
        policy_loss_body(blk, sm + G.off_loss, p.mode, outs, LDO, p.log_var, p.actions, p.ld_act, p.behave,
                         p.ld_beh, p.ref, p.ld_ref, p.adv, (long)J.rows, p.A, ctrl, p.g_surr, p.g_kl,
                         p.row_partials);
    } else if (J.loss == SMX_EPOCH_LOSS_VALUE && tid < 64) {
        // squared error of the 16 rows (ppo.py:323-332): dz3 and the block's mergeable moments of
        // d = ret - V and of ret (explained variance), as value_loss_body forms them per 256 rows
        const ValArgs& q = G.vl;
        const bool ok = tid < nrows;
        const float v = ok ? outs[tid * LDO] : 0.f, g = ok ? q.returns[row0 + tid] : 0.f;
        const float d = g - v, e = v - g;
        if (ok) q.v_dz3[row0 + tid] = (2.0f * e) / (float)G.n_total;
        const float cnt = (float)nrows;
        const float md = smx_wave_sum(ok ? d : 0.f) / cnt;
        const float mg = smx_wave_sum(ok ? g : 0.f) / cnt;
        const float m2d = smx_wave_sum(ok ? (d - md) * (d - md) : 0.f);
        const float m2g = smx_wave_sum(ok ? (g - mg) * (g - mg) : 0.f);
        const float sq = smx_wave_sum(ok ? e * e : 0.f);
        if (tid == 0) {
            float* P = q.v_partials + (size_t)blk * 8;
            P[0] = cnt; P[1] = md; P[2] = m2d; P[3] = mg; P[4] = m2g; P[5] = sq; P[6] = 0.f; P[7] = 0.f;
            if (blk == 0 && q.will_update) ctrl->adam_step_critic += 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward (data gradients).  accT[g] (16 features x 16 rows) += Wt[16 t_g.., :K] . dzT[:K, 16 rows]
// with Wt(m, k) = W[k, m] (W row-major [K, M]): the A operand of step r of a 16-wide K chunk c is
// W[(16c + 4kq + r) * M + 16 t + fm] -- four 4-byte loads per chunk and tile, each a 64-byte run per
// kq group.  k >= K is past the matrix (0); features >= M are pointed out of range.
// ---------------------------------------------------------------------------------------------
template <int NT>
struct TFrag {
    float a[NT][4];
};

template <int NT>
__device__ __forceinline__ void ld_tfrag(TFrag<NT>& f, rsrc_t rw, const unsigned (&wo)[TG], unsigned kstep,
                                         int c, int nch) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const unsigned o = (c < nch) ? wo[g] + (unsigned)c * 16u * kstep : OOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) f.a[g][r] = ld4(rw, (c < nch) ? o + (unsigned)r * kstep : OOB);
    }
}

template <int NT>
__device__ __forceinline__ void mma_tchunk(f32x4 (&acc)[TG], const TFrag<NT>& f, const float* bp) {
    const float4 b = *(const float4*)(bp);
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.a[g][0], b.x, acc[g]);
        acc[g] = MFMA16(f.a[g][1], b.y, acc[g]);
        acc[g] = MFMA16(f.a[g][2], b.z, acc[g]);
        acc[g] = MFMA16(f.a[g][3], b.w, acc[g]);
    }
}

// W row-major [K, M]; tiles t0, t0 + tstep, ... of the M axis
template <int NT>
__device__ __forceinline__ void bwd_tiles(f32x4 (&acc)[TG], rsrc_t rw, int M, int K, const float* in_lds,
                                          int ldi, int t0, int tstep, int fm, int kq) {
    const int nch = (K + 15) >> 4;
    const unsigned kstep = (unsigned)M * 4u;                 // bytes between consecutive k
    unsigned wo[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g) {
        const int f = 16 * (t0 + tstep * g) + fm;
        wo[g] = (g < NT && f < M) ? (unsigned)f * 4u + 4u * kq * kstep : OOB;
    }
    const float* bp = in_lds + fm * ldi + 4 * kq;
    TFrag<NT> P, Q;
    ld_tfrag<NT>(P, rw, wo, kstep, 0, nch);
    for (int c = 0; c < nch; c += 2) {
        ld_tfrag<NT>(Q, rw, wo, kstep, c + 1, nch);
        mma_tchunk<NT>(acc, P, bp + 16 * c);
        ld_tfrag<NT>(P, rw, wo, kstep, c + 2, nch);
        if (c + 1 < nch) mma_tchunk<NT>(acc, Q, bp + 16 * (c + 1));
    }
}

__device__ __forceinline__ void bwd_tiles_n(int nt, f32x4 (&acc)[TG], rsrc_t rw, int M, int K,
                                            const float* in_lds, int ldi, int t0, int tstep, int fm, int kq) {
    switch (nt) {
        case 1: bwd_tiles<1>(acc, rw, M, K, in_lds, ldi, t0, tstep, fm, kq); break;
        case 2: bwd_tiles<2>(acc, rw, M, K, in_lds, ldi, t0, tstep, fm, kq); break;
        case 3: bwd_tiles<3>(acc, rw, M, K, in_lds, ldi, t0, tstep, fm, kq); break;
        case 4: bwd_tiles<4>(acc, rw, M, K, in_lds, ldi, t0, tstep, fm, kq); break;
        case 5: bwd_tiles<5>(acc, rw, M, K, in_lds, ldi, t0, tstep, fm, kq); break;
        default: break;
    }
}

// dzT tiles of one layer: (Wt . dz_up) * relu'(h); the tiles first, first + step, ... < tiles go
// round-robin over the waves.  Results to LDS [row][feature] (when out_lds) and HBM [feature][row]
// (when outT).
__device__ __forceinline__ void bwd_layer(const float* W, int M, int K, const float* in_lds, int ldi,
                                          const float* hT, float* out_lds, int ldo, float* outT, long ldT,
                                          long row0, int nrows, int first, int step, int wv, int fm, int kq) {
    const int tiles = (M + 15) >> 4;
    const rsrc_t rw = make_rsrc(W, (unsigned)K * (unsigned)M * 4u);
    for (int tb = first; tb < tiles; tb += step * NWV * TG) {
        const int t0 = tb + step * wv, ts = step * NWV;
        int nt = (tiles - t0 + ts - 1) / ts;
        nt = nt < 0 ? 0 : (nt > TG ? TG : nt);
        float mk[TG][4];            // ReLU masks, requested in front of the main loop
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            const int f0 = 16 * (t0 + ts * g) + 4 * kq;
            const bool ok = g < nt && f0 < M && fm < nrows;
            const float* q = hT + (size_t)(ok ? f0 : 0) * ldT + row0 + (ok ? fm : 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[g][r] = ok ? q[(size_t)r * ldT] : 0.f;
        }
        f32x4 acc[TG];
#pragma unroll
        for (int g = 0; g < TG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bwd_tiles_n(nt, acc, rw, M, K, in_lds, ldi, t0, ts, fm, kq);
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            if (g < nt) {
                const int f0 = 16 * (t0 + ts * g) + 4 * kq;
                float4 v;
                v.x = (mk[g][0] > 0.f) ? acc[g][0] : 0.f;
                v.y = (mk[g][1] > 0.f) ? acc[g][1] : 0.f;
                v.z = (mk[g][2] > 0.f) ? acc[g][2] : 0.f;
                v.w = (mk[g][3] > 0.f) ? acc[g][3] : 0.f;
                if (out_lds) *(float4*)(out_lds + fm * ldo + f0) = v;
                if (outT && f0 < M && fm < nrows) {
                    float* q = outT + (size_t)f0 * ldT + row0 + fm;
                    q[0] = v.x; q[ldT] = v.y; q[2 * ldT] = v.z; q[3 * ldT] = v.w;
                }
            }
        }
    }
}

__global__ __launch_bounds__(NTH) void epoch_bwd_kernel(EArgs G, smx_ppo_ctrl_t* __restrict__ ctrl) {
    extern __shared__ float sm[];
    __shared__ float S[8 + 2 * MAX_A];
    const EJob J = select_job(G, (int)blockIdx.x);
    const int fs = G.fsplit;
    const int wg = blockIdx.x - J.blk_base;
    const int blk = wg / fs, half = wg - blk * fs;
    const long row0 = (long)blk * ER;
    int nrows = J.rows - (int)row0;
    if (nrows > ER) nrows = ER;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 15, kq = lane >> 4;
    float* dz3s = sm;                       // [16][LDO]
    float* dz2s = sm + G.off_h2;            // [16][ldh2]
    const int ldh2 = G.ldh2;

    for (int idx = tid; idx < (G.off_red >> 2); idx += NTH) *(float4*)(sm + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);
    // the early-exit flag may be raised by workgroup 0 of THIS launch while others start: one lane
    // reads it and the workgroup takes one decision
    if (tid == 0) S[0] = (J.loss == SMX_EPOCH_LOSS_POLICY && ctrl->stop_flag) ? 1.f : 0.f;
    __syncthreads();
    const bool stopped = S[0] != 0.f;
    __syncthreads();
    if (stopped) return;

    if (J.loss == SMX_EPOCH_LOSS_POLICY) {
        const PolArgs& p = G.pl;
        const int A = p.A;
        const int nblk = (J.rows + ER - 1) / ER;
        reduce_row_partials(p.row_partials, nblk, 8 + 2 * A, S, sm + G.off_loss);   // ends with a barrier
        const float n = (float)G.n_total;
        float c_kl, loss;
        loss_and_kl_coef(p.mode, S, n, ctrl, loss, c_kl);
        const float inv_n = 1.0f / n;
        // the early exit taken by THIS pass: workgroup 0 raises the flag below while the others may
        // or may not have read it yet, so every workgroup takes the decision itself
        const bool stop_now = p.check_stop && (double)(S[2] / n) > 4.0 * (double)ctrl->kl_target;
        if (wg == 0) {
            for (int a = tid; a < A; a += NTH) p.dlogvar[a] = (S[8 + a] + c_kl * S[8 + A + a]) * inv_n;
            if (tid == 0)
                write_policy_scalars(S, n, loss, c_kl, p.log_var, A, ctrl, p.check_stop, p.will_update,
                                     p.dlogvar_sumsq, p.stats);
        }
        if (stop_now || !p.will_update) return;
        for (int idx = tid; idx < ER * A; idx += NTH) {
            const int a = idx / ER, nn = idx - a * ER;
            float v = 0.f;
            if (nn < nrows) {
                const size_t i = (size_t)(row0 + nn) * A + a;
                v = (p.g_surr[i] + c_kl * p.g_kl[i]) * inv_n;
                if (half == 0 && J.dz3T) J.dz3T[(size_t)a * J.ldT + row0 + nn] = v;
            }
            dz3s[nn * LDO + a] = v;
        }
    } else {
        if (tid < ER) dz3s[tid * LDO] = (tid < nrows) ? J.dz3[row0 + tid] : 0.f;
    }
    __syncthreads();
    // dz2 = (dz3 . W3) * relu'(h2): all tiles in every workgroup of the row block (cheap), only
    // the first one stores the transposed copy
    bwd_layer(J.W3, J.H2, J.OUT, dz3s, LDO, J.h2T, dz2s, ldh2, half == 0 ? J.dz2T : nullptr, J.ldT, row0,
              nrows, 0, 1, wv, fm, kq);
    __syncthreads();
    // dz1 = (dz2 . W2) * relu'(h1): the feature tiles half, half + fs, ... of this workgroup
    bwd_layer(J.W2, J.H1, J.H2, dz2s, ldh2, J.h1T, nullptr, 0, J.dz1T, J.ldT, row0, nrows, half, fs, wv, fm,
              kq);
}

inline int r32(int v) { return (v + 31) & ~31; }

}  // namespace

extern "C" int32_t smx_epoch_blocks(int64_t rows) { return (int32_t)((rows + ER - 1) / ER); }

extern "C" int32_t smx_epoch_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    return D > 0 && H1 > 0 && H2 > 0 && OUT > 0 && D % 4 == 0 && H1 % 4 == 0 && H2 % 4 == 0 && OUT <= 32 &&
           D <= 2048 && H1 <= 16 * NWV * TG * 2 && H2 <= 16 * NWV * TG * 2;
}

static int fill_args(EArgs& G, const smx_epoch_job_t* jobs, int32_t njobs, const smx_ppo_losses_t* loss,
                     int64_t n_total, int fsplit, bool backward) {
    SMX_REQUIRE(jobs, SMX_E_NULL);
    SMX_REQUIRE(njobs >= 1 && njobs <= MAX_EJOBS, SMX_E_SHAPE);
    memset(&G, 0, sizeof(G));
    G.n = njobs;
    G.n_total = n_total;
    G.fsplit = fsplit;
    int base = 0, maxD = 0, maxH1 = 0, maxH2 = 0, A = 0;
    for (int k = 0; k < njobs; ++k) {
        const smx_epoch_job_t& s = jobs[k];
        SMX_REQUIRE(s.net && s.x, SMX_E_NULL);
        const smx_mlp3_t& n = *s.net;
        SMX_REQUIRE(smx_epoch_supported(n.D, n.H1, n.H2, n.OUT), SMX_E_UNSUPPORTED);
        SMX_REQUIRE(s.rows > 0 && s.rows < (1 << 30), SMX_E_SHAPE);
        SMX_REQUIRE(((uintptr_t)s.x & 15) == 0 && ((uintptr_t)n.W1 & 15) == 0 && ((uintptr_t)n.W2 & 15) == 0 &&
                        ((uintptr_t)n.W3 & 15) == 0 && ((uintptr_t)n.b1 & 15) == 0 && ((uintptr_t)n.b2 & 15) == 0,
                    SMX_E_ALIGN);
        SMX_REQUIRE((!s.h1T && !s.h2T) || (s.h1T && s.h2T && s.ldT >= s.rows), SMX_E_SHAPE);
        EJob& J = G.j[k];
        J.W1 = n.W1; J.b1 = n.b1; J.W2 = n.W2; J.b2 = n.b2; J.W3 = n.W3; J.b3 = n.b3;
        J.D = n.D; J.H1 = n.H1; J.H2 = n.H2; J.OUT = n.OUT;
        J.x = s.x; J.rows = (int)s.rows; J.h1T = s.h1T; J.h2T = s.h2T; J.ldT = (long)s.ldT;
        J.out = s.out; J.out_ld = s.out_ld ? s.out_ld : n.OUT; J.out_act = s.out_act; J.loss = s.loss;
        J.stop = s.stop_flag;
        J.dz3 = s.dz3; J.dz3T = s.dz3T; J.dz2T = s.dz2T; J.dz1T = s.dz1T;
        J.blk_base = base;
        base += smx_epoch_blocks(s.rows) * fsplit;
        if (s.loss == SMX_EPOCH_LOSS_POLICY) {
            SMX_REQUIRE(loss, SMX_E_NULL);
            SMX_REQUIRE(loss->A == n.OUT && loss->A <= MAX_A, SMX_E_SHAPE);
            SMX_REQUIRE(loss->rows == s.rows, SMX_E_SHAPE);
            A = loss->A;
        }
        if (s.loss == SMX_EPOCH_LOSS_VALUE) {
            SMX_REQUIRE(loss && n.OUT == 1, SMX_E_SHAPE);
            SMX_REQUIRE(backward ? s.dz3 != nullptr : (loss->returns && loss->v_dz3 && loss->v_partials), SMX_E_NULL);
        }
        if (backward) SMX_REQUIRE(s.h1T && s.h2T && s.dz2T && s.dz1T, SMX_E_NULL);
        maxD = n.D > maxD ? n.D : maxD; maxH1 = n.H1 > maxH1 ? n.H1 : maxH1; maxH2 = n.H2 > maxH2 ? n.H2 : maxH2;
    }
    if (loss) {
        const smx_ppo_losses_t& a = *loss;
        G.pl.mode = a.mode; G.pl.A = a.A; G.pl.log_var = a.log_var; G.pl.actions = a.actions; G.pl.behave = a.behave;
        G.pl.ref = a.ref; G.pl.adv = a.adv; G.pl.ld_act = a.ld_act; G.pl.ld_beh = a.ld_beh; G.pl.ld_ref = a.ld_ref;
        G.pl.check_stop = a.check_stop; G.pl.will_update = a.will_update; G.pl.g_surr = a.g_surr; G.pl.g_kl = a.g_kl;
        G.pl.row_partials = a.row_partials; G.pl.dlogvar = a.dlogvar; G.pl.dlogvar_sumsq = a.dlogvar_sumsq;
        G.pl.stats = a.stats;
        G.vl.returns = a.returns; G.vl.v_dz3 = a.v_dz3; G.vl.v_partials = a.v_partials;
        G.vl.will_update = a.v_will_update;
    }
    // LDS carve-up: [x tile | h1 tile | h2 tile | out tile | K-split partials | loss scratch]
    G.ldx = r32(maxD) + 4; G.ldh1 = r32(maxH1) + 4; G.ldh2 = r32(maxH2) + 4;
    if (backward) { G.ldx = 0; G.ldh1 = 0; }     // dz3 tile (at 0, stride LDO) | dz2 tile
    G.off_h1 = backward ? ER * LDO : ER * G.ldx;
    G.off_h2 = G.off_h1 + ER * G.ldh1;
    G.off_out = G.off_h2 + ER * G.ldh2;
    G.off_red = G.off_out + (backward ? 0 : ER * LDO);
    G.off_loss = G.off_red + (backward ? 0 : NWV * 2 * 256);
    const int loss_floats = backward ? FIN_CH * (8 + 2 * MAX_A) : ER * (8 * (A ? A : 1) + 1);
    return (G.off_loss + loss_floats) * (int)sizeof(float);
}

extern "C" int smx_epoch_forward_f32(const smx_epoch_job_t* jobs, int32_t njobs, const smx_ppo_losses_t* loss,
                                     smx_ppo_ctrl_t* ctrl, int64_t n_total, smx_stream_t stream) {
    EArgs G;
    const int lds = fill_args(G, jobs, njobs, loss, n_total, 1, false);
    if (lds < 0) return lds;
    SMX_REQUIRE(ctrl, SMX_E_NULL);
    if (loss && loss->mode != SMX_PPO_CLIP && loss->mode != SMX_PPO_ADAPT) return SMX_E_UNSUPPORTED;
    for (int k = 0; k < njobs; ++k)
        if (jobs[k].loss == SMX_EPOCH_LOSS_POLICY)
            SMX_REQUIRE(loss->log_var && loss->actions && loss->behave && loss->ref && loss->adv && loss->g_surr &&
                            loss->g_kl && loss->row_partials, SMX_E_NULL);
    const EJob& Lj = G.j[njobs - 1];
    const int blocks = Lj.blk_base + smx_epoch_blocks(Lj.rows);
    SMX_REQUIRE(lds <= 160 * 1024, SMX_E_UNSUPPORTED);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)epoch_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(epoch_fwd_kernel, dim3(blocks), dim3(NTH), lds, smx_s(stream), G, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_epoch_backward_f32(const smx_epoch_job_t* jobs, int32_t njobs, const smx_ppo_losses_t* loss,
                                      smx_ppo_ctrl_t* ctrl, int64_t n_total, smx_stream_t stream) {
    EArgs G;
    // two workgroups per row block split the dz1 feature tiles; a forward-only pass (no update:
    // statistics and the early-exit decision only) is ONE workgroup
    bool update = true;
    for (int k = 0; k < njobs && jobs; ++k)
        if (jobs[k].loss == SMX_EPOCH_LOSS_POLICY && loss && !loss->will_update) update = false;
    const int fs = 2;
    const int lds = fill_args(G, jobs, njobs, loss, n_total, fs, true);
    if (lds < 0) return lds;
    SMX_REQUIRE(ctrl, SMX_E_NULL);
    for (int k = 0; k < njobs; ++k)
        if (jobs[k].loss == SMX_EPOCH_LOSS_POLICY) {
            SMX_REQUIRE(loss->log_var && loss->g_surr && loss->g_kl && loss->row_partials && loss->dlogvar &&
                            loss->stats, SMX_E_NULL);
            SMX_REQUIRE(!update || jobs[k].dz3T, SMX_E_NULL);
        }
    const EJob& Lj = G.j[njobs - 1];
    int blocks = Lj.blk_base + smx_epoch_blocks(Lj.rows) * fs;
    if (!update) {
        SMX_REQUIRE(njobs == 1, SMX_E_SHAPE);
        blocks = 1;
    }
    hipLaunchKernelGGL(epoch_bwd_kernel, dim3(blocks), dim3(NTH), lds, smx_s(stream), G, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
