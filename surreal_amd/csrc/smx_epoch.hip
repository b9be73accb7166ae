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
//                      L2 -> registers from a copy PACKED in fragment order (one contiguous KB per
//                      load instruction; the optimiser step keeps the copy current), activations go
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
// What shaped the code (all measured, scripts/bench_epoch.py + scripts/micro/): guards are out-of-range
// buffer offsets, never branches (a load under a lane mask is waited for at the end of the masked
// region -- twenty bias words one after the other cost 3000 cycles per layer); nothing conditional
// sits between MFMAs on live accumulators (hipcc then copies them out of and into the MFMA registers
// every chunk); a phase's inputs are requested in one batch at kernel start.
//
// The weight gradients (sums over ALL rows) stay a GEMM launch (smx_gemm.hip), clip-norm + Adam
// one more: 4 dependent launches per epoch instead of 9.
#include "smx_common.h"
#include <stdlib.h>
#include <string.h>

// workgroup barrier for data exchanged through LDS: does NOT wait for the wave's global stores
#define SMX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace {
#include "smx_ppo_loss.inc.h"
#include "smx_epoch_pack.inc.h"
#include "smx_epoch_mma.inc.h"

constexpr int TB = 3;             // the same, backward
constexpr int MAX_EJOBS = 4;
constexpr int XV = 7;             // 16-byte words of its x row a thread fetches up front (D <= 448; more: a loop)
constexpr int LIN = 6;            // loss-input words a thread fetches up front: 16 rows x (5A + 1) <= 6 x 256
constexpr int LINF = 3;           // the same in the 512-thread forward kernels
constexpr int EXCLUSIVE_LDS = 84 * 1024;
constexpr int LDZ = 68;           // row stride of the dz3 tile in LDS (backward: 64 zero-padded columns)
constexpr int LDO = 36;           // row stride of the output tile in LDS (<= 32 outputs)
static_assert(ER == LOSS_ROWS_PER_BLOCK, "a row block is a loss block");
static_assert(NTH == 256, "the shared loss code strides by 256 threads");

// Phase timestamps (cycle counter of thread 0 of every workgroup into a caller-supplied buffer) exist
// only in a build with -DSMX_EPOCH_TIMING (scripts/bench_epoch.py); the product build has none.
#ifdef SMX_EPOCH_TIMING
#define TSTAMP(i) do { if (G.tbuf && threadIdx.x == 0) G.tbuf[(size_t)ts_blk * 32 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
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
    const float *P1, *P2, *P3, *P2T, *P3T;     // packed weights (smx_epoch_pack_f32): layers 1-3, transposed 2-3
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
    int off_g;         // forward + backward in one launch: [dz3 tile (surrogate) | dz3 tile (KL) | dz2 tile]
    int* sync;         // ... and the counter its actor workgroups publish their loss partial rows through
    int nb_pol;        // ... row blocks of its policy job (0: none)
    const int* pol_stop;   // ... and that job's early-exit flag (its workgroups return at once when it is up)
    unsigned long long* kl_slots;   // ... [nb_pol]: a row block's KL sum | 1 << 32, zero on entry
    int fsplit;
    int xsplit;        // forward kernels, two jobs of xsplit row blocks each: job 0 on XCDs 0-3, job 1 on XCDs 4-7 (0: as dispatched)
    long long* tbuf;   // SMX_EPOCH_TIMING builds: per-workgroup phase timestamps (else null)
};

// Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2.  With the actor's row blocks first and the
// critic's behind them, every XCD ran eight of each and pulled BOTH networks' packed weights (rewritten by the optimiser
// launch, so every epoch's first touch is a trip to the memory side) into its L2: 2.2 MB per XCD and launch where 1.1 would do.
// Two jobs of equal size: job 0's row blocks on XCDs 0-3, job 1's on XCDs 4-7.
__device__ __forceinline__ int xcd_job_order(const EArgs& G, int raw) {
    const int nb = G.xsplit;
    if (nb == 0) return raw;
    const int xcd = raw & 7, slot = raw >> 3;
    return (xcd >> 2) * nb + slot * 4 + (xcd & 3);
}

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

// EIGHT wavefronts (two per SIMD) carry the layers: the K loops have no barrier inside, so one wave's weight loads and
// epilogue hide under its SIMD partner's MFMAs (measured on the same loop in the rollout kernel: 42.6 -> 39.8 k cycles
// per pass; a tile's arithmetic does not depend on which wave carries it).  The prologue and the loss are written for
// 256 threads: waves 4-7 only take part in the layers and retire before the loss (a retired wave is not waited for
// by s_barrier).
constexpr int FNWV = 8;
constexpr int FNTH = 64 * FNWV;
constexpr int FTG = 3;            // feature tiles a wave carries per pass (the register budget of two waves per SIMD)

// batch sums S[0 .. stride) of the loss partial rows, 512 threads: staged through LDS with coalesced DEVICE-SCOPE loads
// (the rows were written by other workgroups of this launch, see partial_store), four independent chains per column
// added in a fixed order (the same in every workgroup, whatever the schedule)
__device__ __forceinline__ void fb_reduce_partials(const float* __restrict__ partials, int nblk, int stride, float* S,
                                                   float* buf) {
    const int tid = threadIdx.x;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int b0 = 0; b0 < nblk; b0 += FIN_CH) {
        const int nb = min(FIN_CH, nblk - b0);
        const int cnt = nb * stride;
        const float* src = partials + (size_t)b0 * stride;
        for (int i0 = tid; i0 < cnt; i0 += 8 * FNTH) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = __hip_atomic_load(src + min(i0 + FNTH * u, cnt - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + FNTH * u < cnt) buf[i0 + FNTH * u] = v[u];
        }
        SMX_LDS_BARRIER();
        if (tid < stride) {
            int b = 0;
            for (; b + 4 <= nb; b += 4) {
                t0 += buf[b * stride + tid];
                t1 += buf[(b + 1) * stride + tid];
                t2 += buf[(b + 2) * stride + tid];
                t3 += buf[(b + 3) * stride + tid];
            }
            for (; b < nb; ++b) t0 += buf[b * stride + tid];
        }
        SMX_LDS_BARRIER();
    }
    if (tid < stride) S[tid] = (t0 + t1) + (t2 + t3);
    SMX_LDS_BARRIER();
}

// Forward of up to four jobs' row blocks (FB = false: epoch_fwd_kernel), or forward + loss + data gradients of the
// same rows in ONE launch (FB = true: epoch_fb_kernel, see there).
template <bool FB>
__device__ __forceinline__ void epoch_fwd_body(const EArgs& G, smx_ppo_ctrl_t* __restrict__ ctrl) {
    extern __shared__ float sm[];
    const int bid = xcd_job_order(G, (int)blockIdx.x);
    const int ts_blk = bid;
    (void)ts_blk;
    TSTAMP(0);
    const EJob J = select_job(G, bid);
    const int stopv = J.stop ? __builtin_nontemporal_load(J.stop) : 0;
    const int blk = bid - J.blk_base;
    const long row0 = (long)blk * ER;
    int nrows = J.rows - (int)row0;
    if (nrows > ER) nrows = ER;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 15, kq = lane >> 4;
    TSTAMP(12);

    float* xs = sm;
    float* h1s = sm + G.off_h1;
    float* h2s = sm + G.off_h2;
    float* outs = sm + G.off_out;
    const int ldx = G.ldx, ldh1 = G.ldh1, ldh2 = G.ldh2;

    // ---- everything the workgroup reads that does not depend on its own results is requested
    // HERE, in one batch: the x tile and the loss inputs of its rows (written by earlier launches,
    // possibly on other XCDs: a first touch is a trip to the memory-side cache, not an L2 hit)
    const rsrc_t rx = make_rsrc(J.x, (unsigned)J.rows * (unsigned)J.D * 4u);
    const bool lo = tid < NTH;                           // the first four waves: prologue and loss
    const int xr = (tid & (NTH - 1)) >> 4, xj = tid & 15;   // 16 threads per row
    const unsigned xrow = (unsigned)(row0 + xr) * (unsigned)J.D * 4u;
    const bool xvec = (J.D & 3) == 0;
    float4 xv[XV];
    if (xvec) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int k = 4 * (xj + 16 * i);
            xv[i] = ld16(rx, (lo && k < J.D && xr < nrows) ? xrow + 4u * k : OOB);
        }
    }
    // loss inputs of the 16 rows -> LDS: policy [actions A | behave 2A | ref 2A | adv 1] per row
    const PolArgs& pa = G.pl;
    const int LW = 5 * pa.A + 1;
    float lin[LINF];
    float vret = 0.f;
    // (nothing touches a loaded word before it goes to LDS: a select right behind the load -- "0 for rows past the
    // batch" -- makes hipcc wait for each word in turn; the row test is kept as a bit.  All eight waves fetch, three words
    // each, and (row, column) of word tid + 512 i follows from that of word tid by addition: one integer division per
    // thread, not one per word -- the address arithmetic was 3 k cycles of the actor workgroups' prologue)
    unsigned lin_ok = 0u;
    auto loss_word = [&](int n, int c, bool& ok) -> float {
        const int A = pa.A;
        ok = n < nrows;
        n = ok ? n : 0;
        const long gr = row0 + n;
        const float* q = c < A ? pa.actions + gr * pa.ld_act + c
                       : c < 3 * A ? pa.behave + gr * pa.ld_beh + (c - A)
                       : c < 5 * A ? pa.ref + gr * pa.ld_ref + (c - 3 * A) : pa.adv + gr;
        return *q;
    };
    if (J.loss == SMX_EPOCH_LOSS_POLICY) {
        const int lq = FNTH / LW, lr = FNTH - lq * LW;
        int n = tid / LW, c = tid - n * LW;
#pragma unroll
        for (int i = 0; i < LINF; ++i) {
            const bool in = tid + FNTH * i < ER * LW;
            bool ok;
            lin[i] = loss_word(in ? n : 0, in ? c : 0, ok);
            lin_ok |= (ok && in) ? (1u << i) : 0u;
            n += lq;
            c += lr;
            if (c >= LW) { c -= LW; ++n; }
        }
    } else if (lo && J.loss == SMX_EPOCH_LOSS_VALUE && tid < nrows) {
        vret = G.vl.returns[row0 + tid];
    }
    TSTAMP(13);
    // ---- x tile -> LDS (rows past the batch and k >= D are zero), hidden tiles cleared -------
    for (int idx = tid; idx < (G.off_red - G.off_h1) >> 2; idx += FNTH)
        *(float4*)(h1s + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);
    TSTAMP(14);
    if (xvec && lo) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int k = 4 * (xj + 16 * i);
            if (k < ldx) *(float4*)(xs + xr * ldx + k) = xv[i];
        }
        for (int k = 4 * (xj + 16 * XV); k < ldx; k += 64) {        // D > 64 XV: the rest of the row
            const float4 v = ld16(rx, (k < J.D && xr < nrows) ? xrow + 4u * k : OOB);
            *(float4*)(xs + xr * ldx + k) = v;
        }
    } else if (!xvec) {   // rows of x are only 4-byte aligned (e.g. D = 17)
        for (int idx = tid; idx < ER * ldx; idx += FNTH) {
            const int n = idx / ldx, j = idx - n * ldx;
            xs[idx] = (j < J.D && n < nrows) ? J.x[(size_t)(row0 + n) * J.D + j] : 0.f;
        }
    }
    TSTAMP(15);
    float* lin_s = sm + G.off_loss + loss_scratch_floats(pa.A);      // behind the loss body's own scratch
    if (J.loss == SMX_EPOCH_LOSS_POLICY) {
#pragma unroll
        for (int i = 0; i < LINF; ++i) {
            const int idx = tid + FNTH * i;
            if (idx < ER * LW) lin_s[idx] = ((lin_ok >> i) & 1u) ? lin[i] : 0.f;
        }
        for (int idx = tid + FNTH * LINF; idx < ER * LW; idx += FNTH) {                              // A > 19
            const int n = idx / LW;
            bool ok;
            const float v = loss_word(n, idx - n * LW, ok);
            lin_s[idx] = ok ? v : 0.f;
        }
    }
    if (__builtin_amdgcn_readfirstlane(stopv) != 0) return;
    TSTAMP(1);
    SMX_LDS_BARRIER();
    TSTAMP(2);

    // ---- the three layers: one loop body.  Every wave takes the feature tiles wv, wv + NWV, ...;
    // bias + activation in the fragment; a hidden tile goes to LDS as [row][feature] (the next
    // layer's B operand) and to HBM as [feature][row] (the weight-gradient GEMM's K-contiguous
    // operand and the ReLU mask of the backward kernel), the output tile to LDS (the loss reads it)
    // and to HBM row-major
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        const float* Wp = l == 0 ? J.P1 : (l == 1 ? J.P2 : J.P3);
        const float* bias = l == 0 ? J.b1 : (l == 1 ? J.b2 : J.b3);
        const int H = l == 0 ? J.H1 : (l == 1 ? J.H2 : J.OUT);
        const int K = l == 0 ? J.D : (l == 1 ? J.H1 : J.H2);
        const float* in_lds = l == 0 ? xs : (l == 1 ? h1s : h2s);
        const int ldi = l == 0 ? ldx : (l == 1 ? ldh1 : ldh2);
        float* out_lds = l == 0 ? h1s : (l == 1 ? h2s : outs);
        const int ldo = l == 0 ? ldh1 : (l == 1 ? ldh2 : LDO);
        float* hT = l == 0 ? J.h1T : (l == 1 ? J.h2T : nullptr);
        const int tiles = (H + 15) >> 4;
        const int C2 = (((K + 31) >> 5) + 1) & ~1;
        const rsrc_t rw = make_rsrc(Wp, (unsigned)tiles * (unsigned)C2 * 2048u);
        const rsrc_t rbias = make_rsrc(bias, (unsigned)H * 4u);
        float* stp = l == 2 ? J.out : hT;
        const bool st_ok = stp != nullptr;
        const rsrc_t rst = make_rsrc(stp ? stp : Wp, l == 2 ? (unsigned)J.rows * (unsigned)J.out_ld * 4u
                                                             : (unsigned)H * (unsigned)J.ldT * 4u);
#pragma unroll 1
        for (int tb = 0; tb < tiles; tb += FNWV * FTG) {
            const int t0 = tb + wv;
            int nt = (tiles - t0 + FNWV - 1) / FNWV;
            nt = nt < 0 ? 0 : (nt > FTG ? FTG : nt);
            // the epilogue's bias words, requested in front of the main loop.  Guards are out-of-range
            // buffer offsets, never branches: a load under a lane mask makes hipcc wait for it at the
            // end of the masked region, twenty times in a row
            float bs[FTG][4];
#pragma unroll
            for (int g = 0; g < FTG; ++g) {
                const int f0 = 16 * (t0 + FNWV * g) + 4 * kq;
#pragma unroll
                for (int r = 0; r < 4; ++r) bs[g][r] = ld4(rbias, (g < nt) ? (unsigned)(f0 + r) * 4u : OOB);
            }
            f32x4 acc[TG];
#pragma unroll
            for (int g = 0; g < TG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            TSTAMP(16 + 4 * l);
            if (nt > 2) fwd_tiles<3>(acc, rw, tiles, C2, in_lds, ldi, t0, FNWV, lane);
            else if (nt > 1) fwd_tiles<2>(acc, rw, tiles, C2, in_lds, ldi, t0, FNWV, lane);
            else if (nt > 0) fwd_tiles<1>(acc, rw, tiles, C2, in_lds, ldi, t0, FNWV, lane);
            TSTAMP(17 + 4 * l);
            // hidden tiles: [feature][row] in HBM, lane (fm, kq) holds features f0..f0+3 of row fm;
            // the output tile: row-major.  Stores past the matrix / the batch go to out-of-range offsets.
            const unsigned sstep = l == 2 ? 4u : (unsigned)J.ldT * 4u;                    // bytes between features
            const unsigned srow = l == 2 ? (unsigned)(row0 + fm) * (unsigned)J.out_ld * 4u
                                         : (unsigned)(row0 + fm) * 4u;
#pragma unroll
            for (int g = 0; g < FTG; ++g) {
                if (g < nt) {                                            // wave-uniform
                    const int f0 = 16 * (t0 + FNWV * g) + 4 * kq;    // features f0..f0+3 of data row fm
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float z = acc[g][r] + bs[g][r];
                        // only a wave's FIRST tile can be an output tile (OUT <= 32 < 16 FNWV)
                        if (g == 0 && l == 2) z = act_f(z, J.out_act);
                        else z = (z < 0.f) ? 0.f : z;
                        v[r] = (f0 + r < H) ? z : 0.f;
                    }
                    *(float4*)(out_lds + fm * ldo + f0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = st_ok && fm < nrows && f0 + r < H;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rst,
                                                              ok ? srow + (unsigned)(f0 + r) * sstep : OOB, 0, 0);
                    }
                }
            }
        }
        TSTAMP(18 + 4 * l);
        SMX_LDS_BARRIER();
        TSTAMP(3 + l);
    }

    // ---- the job's loss on the rows it holds (four waves: the other four retire here) ---------
    if constexpr (!FB) {
    if (!lo) return;
    if (J.loss == SMX_EPOCH_LOSS_POLICY) {
        const PolArgs& p = G.pl;
        // inputs staged in LDS: row stride LW, [actions | behave | ref | adv]
        policy_loss_body(blk, sm + G.off_loss, p.mode, outs, LDO, p.log_var, lin_s, LW, lin_s + p.A, LW,
                         lin_s + 3 * p.A, LW, lin_s + 5 * p.A, (long)J.rows, p.A, ctrl, p.g_surr, p.g_kl,
                         p.row_partials, 1.0f, false, nullptr, nullptr, 0, row0, LW);
    } else if (J.loss == SMX_EPOCH_LOSS_VALUE && tid < 64) {
        // squared error of the 16 rows (ppo.py:323-332): dz3 and the block's mergeable moments of
        // d = ret - V and of ret (explained variance), as value_loss_body forms them per 256 rows
        const ValArgs& q = G.vl;
        const bool ok = tid < nrows;
        const float v = ok ? outs[tid * LDO] : 0.f, g = ok ? vret : 0.f;
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
    TSTAMP(6);
    } else {
    // =========================================================================================
    // FB: the loss, then the data gradients of the SAME rows without leaving the workgroup -- the
    // hidden activations (ReLU masks) and the loss's per-element gradient terms are still in LDS.
    //   dz3 = (g_surr + c_kl g_kl) / n needs the batch mean KL (adapt mode: c_kl = beta + 2 eta max(0, KL - 2 kl_target),
    //   ppo.py:272-276), i.e. every actor workgroup's KL sum.  The actor workgroups publish their partial row
    //   (device-scope write-through stores, then one counter increment), and while the others arrive they multiply BOTH
    //   right-hand sides through the output layer (dz2 is linear in dz3: W3^T g_surr / n and W3^T g_kl / n, a K <= 32
    //   product each); only then do they look at the counter, add up the KL column of the partial rows (device-scope
    //   loads; one value per lane + a butterfly: the same bits in every workgroup), combine under the ReLU mask and
    //   carry on with the one expensive product, dz1.  (The first version used agent-scope fences around the counter:
    //   on gfx950 those write back / invalidate the whole L2 -- 4 us per side, and every workgroup of the XCD ran cold
    //   afterwards.)  The critic's workgroups never wait.  In clip mode nothing in the gradient depends on the batch.
    //   The epoch's scalars -- all column sums, statistics, log_var's gradient, the early-exit flag, the step counters
    //   -- are formed ONCE, by the last workgroup of the grid after its own work (a critic workgroup when the launch
    //   carries both jobs: those finish ~8 us ahead of the actor's); the optimiser launch that follows honours the flag.
    // All eight wavefronts stay: the four that do not run the loss execute its barriers.
    // =========================================================================================
    // (static LDS precedes the dynamic region unpadded: its size stays a multiple of 16 bytes, or every 16-byte LDS
    // access of the kernel would sit off its natural alignment)
    __shared__ __attribute__((aligned(16))) float S[8 + 2 * MAX_A + 8];
    float& s_kl = S[8 + 2 * MAX_A];
    int& wait_ok = *(int*)&S[8 + 2 * MAX_A + 1];
    static_assert(sizeof(S) % 16 == 0, "static LDS is a multiple of 16 bytes");
    const PolArgs& p = G.pl;
    const bool policy = J.loss == SMX_EPOCH_LOSS_POLICY;
    const bool adapt = policy && p.mode == SMX_PPO_ADAPT;
    const int A = p.A;
    const int pstride = 8 + 2 * A;
    float* gss = sm + G.off_g;                 // [16][LDZ]: dz3 right-hand side (surrogate share / the critic's dz3)
    float* gks = gss + ER * LDZ;               // [16][LDZ]: the KL share (adapt)
    float* dz2s = gks + ER * LDZ;              // [16][ldh2]
    const float nf = (float)G.n_total;
    const float inv_n = 1.0f / nf;
    // was the actor's early-exit flag up when the launch started?  (its workgroups have returned then; read by the
    // workgroup that forms the epoch's scalars, long before it could raise the flag itself)
    const int pol_stopped = (G.pol_stop && blockIdx.x == gridDim.x - 1) ? __builtin_nontemporal_load(G.pol_stop) : 0;
    // the output layer's transposed weights for this wave's dz2 tiles (chunk 0: K = OUT <= 32), requested BEFORE the
    // loss: the optimiser launch has just rewritten them on other XCDs, a first touch is a trip to the memory-side cache
    const int tiles2 = (J.H2 + 15) >> 4;
    float4 wa[FTG], wb[FTG];
    {
        const rsrc_t rw3 = make_rsrc(J.P3T, (unsigned)tiles2 * (unsigned)pack_chunks(J.OUT) * 2048u);
#pragma unroll
        for (int g = 0; g < FTG; ++g) {
            const int t = wv + FNWV * g;
            const unsigned o = t < tiles2 ? ((unsigned)t * (unsigned)pack_chunks(J.OUT) * 512u + (unsigned)lane * 4u) * 4u : OOB;
            wa[g] = ld16(rw3, o);
            wb[g] = ld16(rw3, t < tiles2 ? o + 1024u : OOB);
        }
    }
    if (policy) {
        if (lo) {
            policy_loss_body<true>(blk, sm + G.off_loss, p.mode, outs, LDO, p.log_var, lin_s, LW, lin_s + p.A, LW,
                                   lin_s + 3 * p.A, LW, lin_s + 5 * p.A, (long)J.rows, p.A, ctrl, nullptr, nullptr,
                                   p.row_partials, 1.0f, false, nullptr, nullptr, 0, row0, LW, G.kl_slots);
        } else {
#pragma unroll
            for (int i = 0; i < POLICY_LOSS_BARRIERS; ++i) SMX_LDS_BARRIER();
        }
    } else if (tid < 64) {
        const ValArgs& q = G.vl;
        const bool ok = tid < nrows;
        const float v = ok ? outs[tid * LDO] : 0.f, g = ok ? vret : 0.f;
        const float d = g - v, e = v - g;
        const float dz = (2.0f * e) / nf;
        if (ok) q.v_dz3[row0 + tid] = dz;
        if (tid < ER) gss[tid * LDZ] = ok ? dz : 0.f;
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
    SMX_LDS_BARRIER();
    TSTAMP(6);
    if (policy) {
        // right-hand side tiles from the loss's scratch, zero padded to the 32 columns the K loop reads
        const float* sc = sm + G.off_loss;
        const float* e_dmu = loss_elem_dmu(sc, A);
        const float* e_dkl = loss_elem_dkl(sc, A);
        const float* r_dll = loss_row_dll(sc, A);
        for (int idx = tid; idx < ER * 32; idx += FNTH) {
            const int n = idx >> 5, a = idx & 31;
            const bool ok = n < nrows && a < A;
            const int i = ok ? n * A + a : 0;
            const float gs = r_dll[ok ? n : 0] * e_dmu[i];
            gss[n * LDZ + a] = ok ? gs * inv_n : 0.f;
            if (adapt) gks[n * LDZ + a] = ok ? e_dkl[i] * inv_n : 0.f;
        }
    }
    SMX_LDS_BARRIER();
    // ---- dz2 (before the mask) for both right-hand sides: every wave's tiles wv, wv + 8, ... (one chunk, K <= 32;
    // the weight fragments are shared by the two right-hand sides) ------------------------------------------
    f32x4 aS[FTG], aK[FTG];
#pragma unroll
    for (int g = 0; g < FTG; ++g) { aS[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; aK[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    {
        const float* bs = gss + (lane & 15) * LDZ + 8 * (lane >> 4);
        const float4 s0 = *(const float4*)bs, s1 = *(const float4*)(bs + 4);
        float4 k0 = make_float4(0.f, 0.f, 0.f, 0.f), k1 = k0;
        if (adapt) {
            const float* bk = gks + (lane & 15) * LDZ + 8 * (lane >> 4);
            k0 = *(const float4*)bk; k1 = *(const float4*)(bk + 4);
        }
#pragma unroll
        for (int g = 0; g < FTG; ++g) {
            if (wv + FNWV * g < tiles2) {                       // wave-uniform; each tile has its own accumulators
                aS[g] = MFMA16(wa[g].x, s0.x, aS[g]); aS[g] = MFMA16(wa[g].y, s0.y, aS[g]);
                aS[g] = MFMA16(wa[g].z, s0.z, aS[g]); aS[g] = MFMA16(wa[g].w, s0.w, aS[g]);
                aS[g] = MFMA16(wb[g].x, s1.x, aS[g]); aS[g] = MFMA16(wb[g].y, s1.y, aS[g]);
                aS[g] = MFMA16(wb[g].z, s1.z, aS[g]); aS[g] = MFMA16(wb[g].w, s1.w, aS[g]);
                if (adapt) {
                    aK[g] = MFMA16(wa[g].x, k0.x, aK[g]); aK[g] = MFMA16(wa[g].y, k0.y, aK[g]);
                    aK[g] = MFMA16(wa[g].z, k0.z, aK[g]); aK[g] = MFMA16(wa[g].w, k0.w, aK[g]);
                    aK[g] = MFMA16(wb[g].x, k1.x, aK[g]); aK[g] = MFMA16(wb[g].y, k1.y, aK[g]);
                    aK[g] = MFMA16(wb[g].z, k1.z, aK[g]); aK[g] = MFMA16(wb[g].w, k1.w, aK[g]);
                }
            }
        }
    }
    TSTAMP(7);
    // ---- the batch KL (adapt: every actor workgroup needs it now) ---------------------------------------
    const int nbp = G.nb_pol;
    auto wait_count = [&]() {        // every actor workgroup of the launch has published its partial row
        if (tid == 0) {
            const long long t0 = (long long)wall_clock64();
            int okw = 1;
            while (__hip_atomic_load(G.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nbp) {
                if ((long long)wall_clock64() - t0 > 25000000LL) { okw = 0; break; }     // 0.25 s (100 MHz): a lost workgroup
                __builtin_amdgcn_s_sleep(1);
            }
            wait_ok = wait_ok & okw;
            if (!okw) atomicOr(&ctrl->reserved[1], 1);
        }
        __syncthreads();
    };
    // The batch KL: lane l of wavefront 0 takes row blocks l, l + 64, ... -- it polls the block's slot (ONE device-scope
    // 8-byte load: the sum and its "there" bit were stored together, so there is no separate flag round trip and the
    // producer never waits for its own store) -- then a butterfly: the same bits in every workgroup.
    auto kl_total = [&]() -> float {
        if (tid < 64) {
            float t = 0.f;
            int okw = 1;
            const long long t0 = (long long)wall_clock64();
            for (int b = tid; b < nbp; b += 64) {
                unsigned long long w;
                while (((w = __hip_atomic_load(G.kl_slots + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0ull) {
                    if ((long long)wall_clock64() - t0 > 25000000LL) { okw = 0; break; }    // 0.25 s (100 MHz): a lost workgroup
                    __builtin_amdgcn_s_sleep(1);
                }
                t += __uint_as_float((unsigned)w);
            }
            t = smx_wave_sum(t);
            okw = __all(okw);
            if (tid == 0) {
                s_kl = t;
                wait_ok = wait_ok & okw;         // (never un-flag an earlier timeout of this workgroup)
                if (!okw) atomicOr(&ctrl->reserved[1], 1);
            }
        }
        SMX_LDS_BARRIER();
        return s_kl;
    };
    auto kl_coef = [&](float klsum, float& ck, bool& stop) {
        float S3[3] = {0.f, 0.f, klsum}, ls;
        loss_and_kl_coef(p.mode, S3, nf, ctrl, ls, ck);
        stop = (p.check_stop && (double)(klsum / nf) > 4.0 * (double)ctrl->kl_target) || wait_ok == 0;
    };
    float c_kl = 0.f;
    bool stop_now = false;
    if (tid == 0) wait_ok = 1;          // (thread 0 is the only writer; every wait of this workgroup ANDs into it)
    if (adapt) kl_coef(kl_total(), c_kl, stop_now);
    // the partial ROW (all column sums: the finalizing workgroup's input) went out with device-scope stores during the
    // loss, several microseconds ago: wavefronts 0 and 1 make sure they have completed, then the counter moves
    if (policy && tid < 128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SMX_LDS_BARRIER();
    if (policy && tid == 0) __hip_atomic_fetch_add(G.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TSTAMP(8);
    if (!stop_now) {
        // ---- dz2 = (W3^T dz3) * relu'(h2): LDS tile (dz1's B operand) + transposed copy (weight gradients) ----
        {
            const rsrc_t rout = make_rsrc(J.dz2T, (unsigned)J.H2 * (unsigned)J.ldT * 4u);
#pragma unroll
            for (int g = 0; g < FTG; ++g) {
                const int t = wv + FNWV * g;
                if (t < tiles2) {                                   // wave-uniform
                    const int f0 = 16 * t + 4 * kq;
                    const float4 m = *(const float4*)(h2s + fm * ldh2 + f0);
                    float4 v;
                    v.x = (m.x > 0.f) ? aS[g][0] + c_kl * aK[g][0] : 0.f;
                    v.y = (m.y > 0.f) ? aS[g][1] + c_kl * aK[g][1] : 0.f;
                    v.z = (m.z > 0.f) ? aS[g][2] + c_kl * aK[g][2] : 0.f;
                    v.w = (m.w > 0.f) ? aS[g][3] + c_kl * aK[g][3] : 0.f;
                    *(float4*)(dz2s + fm * ldh2 + f0) = v;
                    const bool ok = fm < nrows;
                    const unsigned o = ((unsigned)f0 * (unsigned)J.ldT + (unsigned)(row0 + fm)) * 4u, st = (unsigned)J.ldT * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), rout, ok && f0 < J.H2 ? o : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), rout, ok && f0 + 1 < J.H2 ? o + st : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.z), rout, ok && f0 + 2 < J.H2 ? o + 2 * st : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.w), rout, ok && f0 + 3 < J.H2 ? o + 3 * st : OOB, 0, 0);
                }
            }
        }
        if (policy) {       // dz3^T for the output layer's weight gradient
            for (int idx = tid; idx < ER * A; idx += FNTH) {
                const int a = idx / ER, n = idx - a * ER;
                const float v = gss[n * LDZ + a] + c_kl * gks[n * LDZ + a];
                if (n < nrows) J.dz3T[(size_t)a * J.ldT + row0 + n] = v;
            }
        }
        SMX_LDS_BARRIER();
        TSTAMP(9);
        // ---- dz1 = (W2^T dz2) * relu'(h1) -----------------------------------------------------------
        {
            const int tiles1 = (J.H1 + 15) >> 4;
            const int C2h = pack_chunks(J.H2);
            const rsrc_t rw2 = make_rsrc(J.P2T, (unsigned)tiles1 * (unsigned)C2h * 2048u);
            const rsrc_t rout = make_rsrc(J.dz1T, (unsigned)J.H1 * (unsigned)J.ldT * 4u);
#pragma unroll 1
            for (int tb = 0; tb < tiles1; tb += FNWV * FTG) {
                const int t0 = tb + wv;
                int nt = (tiles1 - t0 + FNWV - 1) / FNWV;
                nt = nt < 0 ? 0 : (nt > FTG ? FTG : nt);
                f32x4 acc[TG];
#pragma unroll
                for (int g = 0; g < TG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (nt > 2) fwd_tiles<3>(acc, rw2, tiles1, C2h, dz2s, ldh2, t0, FNWV, lane);
                else fwd_tiles<2>(acc, rw2, tiles1, C2h, dz2s, ldh2, t0, FNWV, lane);
#pragma unroll
                for (int g = 0; g < FTG; ++g) {
                    if (g < nt) {
                        const int f0 = 16 * (t0 + FNWV * g) + 4 * kq;
                        const float4 m = *(const float4*)(h1s + fm * ldh1 + f0);
                        const float vx = (m.x > 0.f) ? acc[g][0] : 0.f, vy = (m.y > 0.f) ? acc[g][1] : 0.f;
                        const float vz = (m.z > 0.f) ? acc[g][2] : 0.f, vw = (m.w > 0.f) ? acc[g][3] : 0.f;
                        const bool ok = fm < nrows;
                        const unsigned o = ((unsigned)f0 * (unsigned)J.ldT + (unsigned)(row0 + fm)) * 4u, st = (unsigned)J.ldT * 4u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vx), rout, ok && f0 < J.H1 ? o : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vy), rout, ok && f0 + 1 < J.H1 ? o + st : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vz), rout, ok && f0 + 2 < J.H1 ? o + 2 * st : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vw), rout, ok && f0 + 3 < J.H1 ? o + 3 * st : OOB, 0, 0);
                    }
                }
            }
        }
    }
    TSTAMP(10);
    // ---- the epoch's scalars: the last workgroup of the grid, once its own rows are done -------------------
    if (nbp > 0 && blockIdx.x == gridDim.x - 1 && pol_stopped == 0) {
        wait_count();
        fb_reduce_partials(p.row_partials, nbp, pstride, S, sm + G.off_loss);
        const float klsum = kl_total();                    // the KL sum every actor workgroup used: the same bits
        float loss = 0.f, ck = 0.f;
        if (tid == 0) S[2] = klsum;
        SMX_LDS_BARRIER();
        loss_and_kl_coef(p.mode, S, nf, ctrl, loss, ck);
        for (int a = tid; a < A; a += FNTH) p.dlogvar[a] = (S[8 + a] + ck * S[8 + A + a]) * inv_n;
        if (tid == 0 && wait_ok)
            write_policy_scalars(S, nf, loss, ck, p.log_var, A, ctrl, p.check_stop, p.will_update, p.dlogvar_sumsq,
                                 p.stats);
    }
    TSTAMP(11);
    }
}

__global__ __launch_bounds__(FNTH) void epoch_fwd_kernel(EArgs G, smx_ppo_ctrl_t* __restrict__ ctrl) {
    epoch_fwd_body<false>(G, ctrl);
}

// One launch per epoch for [forward + loss + data gradients] of the row blocks of both networks: what
// epoch_fwd_kernel + epoch_bwd_kernel do in two (the batch means travel through a counter inside the launch,
// see the FB part of epoch_fwd_body).  The actor workgroups of a launch must be co-resident (<= one per CU,
// dispatched before anything that could wait on them: they are); a wait is bounded all the same.
__global__ __launch_bounds__(FNTH) void epoch_fb_kernel(EArgs G, smx_ppo_ctrl_t* __restrict__ ctrl) {
    epoch_fwd_body<true>(G, ctrl);
}

// ---------------------------------------------------------------------------------------------
// backward (data gradients).  accT[g] (16 features x 16 rows) += Wt[16 t_g.., :K] . dzT[:K, 16 rows]
// with Wt(m, k) = W[k, m]: the packed copy holds the TRANSPOSED matrices of layers 2 and 3 in the same
// fragment order as the forward weights, so this is the forward loop on other operands.  (Read from
// the row-major matrices -- four 4-byte loads per 4 MFMAs -- the dz1 product ran at half its MFMA rate,
// bound by the CU's address unit.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void epoch_bwd_kernel(EArgs G, smx_ppo_ctrl_t* __restrict__ ctrl) {
    extern __shared__ float sm[];
    __shared__ float S[8 + 2 * MAX_A];
    const int ts_blk = blockIdx.x;
    (void)ts_blk;
    TSTAMP(0);
    const EJob J = select_job(G, (int)blockIdx.x);
    const int fs = G.fsplit;
    const int wg = blockIdx.x - J.blk_base;
    const int nb = (J.rows + ER - 1) / ER;
    // workgroup -> (row block, feature split).  Consecutive workgroup ids go round-robin over the 8
    // XCDs: keep row block b on XCD b % 8, where the forward kernel ran it and left its activations
    // and loss tiles in that XCD's L2 (when the job's block count allows the bijection)
    int blk, half;
    if ((nb & 7) == 0 && (J.blk_base & 7) == 0) {
        const int s = wg >> 3;
        half = s % fs;
        blk = (s / fs) * 8 + (wg & 7);
    } else {
        blk = wg / fs;
        half = wg - blk * fs;
    }
    const long row0 = (long)blk * ER;
    int nrows = J.rows - (int)row0;
    if (nrows > ER) nrows = ER;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 15, kq = lane >> 4;
    float* dz3s = sm;                       // [16][LDZ]
    float* dz2s = sm + G.off_h2;            // [16][ldh2]
    const int ldh2 = G.ldh2;
    const bool policy = J.loss == SMX_EPOCH_LOSS_POLICY;
    // data-parallel epochs: the backward pass runs on the two right-hand sides g_surr / n and g_kl / n
    // separately (the loss gradient is linear in dz3; the KL coefficient needs the GLOBAL mean KL and is
    // applied after the all-reduce, smx_ppo_epoch_combine_f32): no batch means, no statistics here
    const bool dp_surr = J.loss == SMX_EPOCH_RHS_SURR, dp_kl = J.loss == SMX_EPOCH_RHS_KL;
    const bool actor = policy || dp_surr || dp_kl;
    const PolArgs& p = G.pl;

    // ---- requested up front, in one batch: this workgroup's share of the loss tiles -----------
    const int A = p.A;
    float gs0 = 0.f, gk0 = 0.f, gs1 = 0.f, gk1 = 0.f, vd = 0.f;       // <= 2 (row, action) pairs per thread
    if (actor) {             // (unconditional loads from clamped addresses: no load under a lane mask)
        const int last = ER * A - 1;
        const int i0 = tid < last ? tid : last, i1 = tid + NTH < last ? tid + NTH : last;
        const int a0 = i0 / ER, n0 = i0 - a0 * ER, a1 = i1 / ER, n1 = i1 - a1 * ER;
        const size_t e0 = (size_t)(row0 + (n0 < nrows ? n0 : 0)) * A + a0;
        const size_t e1 = (size_t)(row0 + (n1 < nrows ? n1 : 0)) * A + a1;
        gs0 = p.g_surr[e0]; gk0 = p.g_kl[e0];
        gs1 = p.g_surr[e1]; gk1 = p.g_kl[e1];
    } else {
        vd = J.dz3[row0 + (tid < nrows ? tid : 0)];
    }
    for (int idx = tid; idx < (G.off_red >> 2); idx += NTH) *(float4*)(sm + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);
    // the early-exit flag may be raised by workgroup 0 of THIS launch while others start: one lane
    // reads it and the workgroup takes one decision
    if (tid == 0) S[0] = (actor && ctrl->stop_flag) ? 1.f : 0.f;
    SMX_LDS_BARRIER();
    const bool stopped = S[0] != 0.f;
    SMX_LDS_BARRIER();
    if (stopped) return;
    TSTAMP(1);

    if (policy) {
        const int nblk = nb;
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
        TSTAMP(2);
        if (stop_now || !p.will_update) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTH * i;
            if (idx < ER * A) {
                const int a = idx / ER, nn = idx - a * ER;
                const float v = nn < nrows ? ((i ? gs1 : gs0) + c_kl * (i ? gk1 : gk0)) * inv_n : 0.f;
                if (nn < nrows && half == 0 && J.dz3T) J.dz3T[(size_t)a * J.ldT + row0 + nn] = v;   // (a store: nothing waits for it)
                dz3s[nn * LDZ + a] = v;
            }
        }
    } else if (actor) {
        const float inv_n = 1.0f / (float)G.n_total;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTH * i;
            if (idx < ER * A) {
                const int a = idx / ER, nn = idx - a * ER;
                const float v = nn < nrows ? (dp_surr ? (i ? gs1 : gs0) : (i ? gk1 : gk0)) * inv_n : 0.f;
                if (nn < nrows && half == 0 && J.dz3T) J.dz3T[(size_t)a * J.ldT + row0 + nn] = v;
                dz3s[nn * LDZ + a] = v;
            }
        }
    } else {
        if (tid < ER) dz3s[tid * LDZ] = tid < nrows ? vd : 0.f;
    }
    SMX_LDS_BARRIER();
    TSTAMP(3);
    // ---- dz2 = (dz3 . W3) * relu'(h2): all tiles in every workgroup of the row block (cheap; only
    // the first one stores the transposed copy); dz1 = (dz2 . W2) * relu'(h1): the feature tiles
    // half, half + fs, ... of this workgroup.  One loop body for both (code size, see above).
#pragma unroll 1
    for (int l = 0; l < 2; ++l) {
        const float* Wp = l == 0 ? J.P3T : J.P2T;
        const int M = l == 0 ? J.H2 : J.H1, K = l == 0 ? J.OUT : J.H2;
        const float* in_lds = l == 0 ? dz3s : dz2s;
        const int ldi = l == 0 ? LDZ : ldh2;
        const float* hT = l == 0 ? J.h2T : J.h1T;
        float* out_lds = l == 0 ? dz2s : nullptr;
        float* outT = l == 0 ? (half == 0 ? J.dz2T : nullptr) : J.dz1T;
        const int first = l == 0 ? 0 : half, step = l == 0 ? 1 : fs;
        const int tiles = (M + 15) >> 4;
        const int C2 = (((K + 31) >> 5) + 1) & ~1;
        const rsrc_t rw = make_rsrc(Wp, (unsigned)tiles * (unsigned)C2 * 2048u);
        const rsrc_t rmask = make_rsrc(hT, (unsigned)M * (unsigned)J.ldT * 4u);
        const rsrc_t rout = make_rsrc(outT ? outT : hT, (unsigned)M * (unsigned)J.ldT * 4u);
#pragma unroll 1
        for (int tb = first; tb < tiles; tb += step * NWV * TB) {
            const int t0 = tb + step * wv, ts = step * NWV;
            int nt = (tiles - t0 + ts - 1) / ts;
            nt = nt < 0 ? 0 : (nt > TB ? TB : nt);
            float mk[TB][4];            // ReLU masks, requested in front of the main loop (guards: out-of-range offsets)
#pragma unroll
            for (int g = 0; g < TB; ++g) {
                const int f0 = 16 * (t0 + ts * g) + 4 * kq;
                const bool ok = g < nt && f0 < M && fm < nrows;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mk[g][r] = ld4(rmask, ok ? ((unsigned)(f0 + r) * (unsigned)J.ldT + (unsigned)(row0 + fm)) * 4u : OOB);
            }
            f32x4 acc[TG];
#pragma unroll
            for (int g = 0; g < TG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            TSTAMP(8 + 4 * l);
            if (nt > 1) fwd_tiles<TB>(acc, rw, tiles, C2, in_lds, ldi, t0, ts, lane);
            else if (nt > 0) fwd_tiles<1>(acc, rw, tiles, C2, in_lds, ldi, t0, ts, lane);
            TSTAMP(9 + 4 * l);
#pragma unroll
            for (int g = 0; g < TB; ++g) {
                if (g < nt) {
                    const int f0 = 16 * (t0 + ts * g) + 4 * kq;
                    float4 v;
                    v.x = (mk[g][0] > 0.f) ? acc[g][0] : 0.f;
                    v.y = (mk[g][1] > 0.f) ? acc[g][1] : 0.f;
                    v.z = (mk[g][2] > 0.f) ? acc[g][2] : 0.f;
                    v.w = (mk[g][3] > 0.f) ? acc[g][3] : 0.f;
                    if (out_lds) *(float4*)(out_lds + fm * ldh2 + f0) = v;
                    const bool ok = outT != nullptr && f0 < M && fm < nrows;
                    const unsigned o = ((unsigned)f0 * (unsigned)J.ldT + (unsigned)(row0 + fm)) * 4u, st = (unsigned)J.ldT * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), rout, ok ? o : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), rout, ok ? o + st : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.z), rout, ok ? o + 2 * st : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.w), rout, ok ? o + 3 * st : OOB, 0, 0);
                }
            }
        }
        TSTAMP(10 + 4 * l);
        SMX_LDS_BARRIER();
        TSTAMP(4 + l);
    }
}

inline int r64(int v) { return (v + 63) & ~63; }

struct PackNet {
    const float *W1, *W2, *W3;
    float* packed;
    int D, H1, H2, OUT;
    long base;         // first 16-byte word of this net in the launch's index space
};
struct PackArgs {
    PackNet n[MAX_EJOBS];
    int count;
    long total;
};

// one thread per 16-byte word of the packed copy: [tile][chunk][half][lane][4] <- X[16 tile + (lane & 15)]
// [32 chunk + 8 (lane >> 4) + 4 half + 0..3], zero outside the matrix; X = W or W^T
__device__ __forceinline__ void pack_word(const PackArgs& P, long i) {
    int pi = 0;
#pragma unroll
    for (int k = 1; k < MAX_EJOBS; ++k) pi += (k < P.count && i >= P.n[k].base) ? 1 : 0;
    const PackNet N = P.n[pi];
    long w = i - N.base;
    int bn = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k) bn += (w >= pack_off(N.D, N.H1, N.H2, N.OUT, k)) ? 1 : 0;
    w -= pack_off(N.D, N.H1, N.H2, N.OUT, bn);
    const float* W = bn == 0 ? N.W1 : ((bn == 1 || bn == 3) ? N.W2 : N.W3);
    const bool tr = bn >= 3;
    const int M = bn == 0 ? N.H1 : (bn == 1 ? N.H2 : (bn == 2 ? N.OUT : (bn == 3 ? N.H1 : N.H2)));
    const int K = bn == 0 ? N.D : (bn == 1 ? N.H1 : (bn == 2 ? N.H2 : (bn == 3 ? N.H2 : N.OUT)));
    const int C2 = pack_chunks(K);
    const int lane = (int)(w & 63), half = (int)((w >> 6) & 1);
    const long tc = w >> 7;
    const int c = (int)(tc % C2), t = (int)(tc / C2);
    const int m = 16 * t + (lane & 15), k = 32 * c + 8 * (lane >> 4) + 4 * half;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (k + r < K) v[r] = tr ? W[(size_t)(k + r) * M + m] : W[(size_t)m * K + k + r];
    }
    *(float4*)(N.packed + 4 * (i - N.base)) = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void epoch_pack_kernel(PackArgs P) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < P.total) pack_word(P, i);
}

// ---------------------------------------------------------------------------------------------
// Everything the epoch loop needs prepared once per learn, in ONE launch instead of seven: the
// z-filtered step-0 observations through the model's filter (+ their transposed copy, the weight-
// gradient GEMMs' K-contiguous operand) and through the reference policy's filter, the z-filtered
// obs_next rows (the critic's tail rows), the reference policy's std columns, and the packed weight
// copies of up to four networks.  Arithmetic as smx_zfilter_forward_f32 / _forward_sums_f32 /
// smx_epoch_pack_f32 (bit-identical; z_filter.py:74-77, builders.py:126-129).
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
    const float* obs0; long ld_obs0; long rows; int D;
    const float *zmean, *zstd;
    float *xn, *xnT; long ldT;
    const float *rsum, *rsumsq, *rcount; float reps; int ref_filter;
    float* xr;
    const float* obs_next; long ld_next; float* xnext;
    const float* ref_log_var; int A; float* ref_std; long ld_ref;
    int* zero_words; int n_zero;
    PackArgs pack;
    long n_x, n_std;            // rows * D, rows * A
};

__device__ __forceinline__ float zclamp_e(float x, float m, float sdev) {
    float v = (x - m) / sdev;   // z_filter.py:77
    if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);       // torch.clamp keeps NaN
    return v;
}

__global__ __launch_bounds__(256) void epoch_prepare_kernel(PrepArgs P) {
    const long tid0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    for (long i = tid0; i < P.n_zero; i += stride) P.zero_words[i] = 0;
    const long nx = P.n_x, n_next = P.xnext ? nx : 0;
    const long total = 2 * nx + n_next + P.n_std + P.pack.total;
    for (long i = tid0; i < total; i += stride) {
        if (i < 2 * nx + n_next) {
            const int which = i < nx ? 0 : (i < 2 * nx ? 1 : 2);      // xn | xr | xnext
            const long e = i - (which == 0 ? 0 : (which == 1 ? nx : 2 * nx));
            const long r = e / P.D;
            const int k = (int)(e - r * P.D);
            const float x = which == 2 ? P.obs_next[r * P.ld_next + k] : P.obs0[r * P.ld_obs0 + k];
            float v = x;
            if (which == 1) {
                if (P.ref_filter) {
                    const float c = P.rcount[0];
                    const float m = P.rsum[k] / c;
                    const float var = P.rsumsq[k] / c - m * m;
                    float sd = sqrtf(var);
                    if (sd == sd) sd = fmaxf(sd, P.reps);
                    v = zclamp_e(x, m, sd);
                }
                P.xr[e] = v;
            } else {
                if (P.zmean) v = zclamp_e(x, P.zmean[k], P.zstd[k]);
                if (which == 0) {
                    P.xn[e] = v;
                    if (P.xnT) P.xnT[(size_t)k * P.ldT + r] = v;
                } else {
                    P.xnext[e] = v;
                }
            }
        } else if (i < 2 * nx + n_next + P.n_std) {
            const long e = i - (2 * nx + n_next);
            const long r = e / P.A;
            const int a = (int)(e - r * P.A);
            P.ref_std[r * P.ld_ref + a] = expf(P.ref_log_var[a]);      // exp(log_var) * ones_like(mean)
        } else {
            pack_word(P.pack, i - (2 * nx + n_next + P.n_std));
        }
    }
}

long long* g_tbuf = nullptr;

}  // namespace

extern "C" int64_t smx_epoch_packed_floats(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    return 4 * pack_off(D, H1, H2, OUT, 5);
}

static int fill_pack(PackArgs& P, const smx_epoch_pack_t* items, int32_t n) {
    SMX_REQUIRE(items, SMX_E_NULL);
    SMX_REQUIRE(n >= 1 && n <= MAX_EJOBS, SMX_E_SHAPE);
    P.count = n;
    long base = 0;
    for (int k = 0; k < n; ++k) {
        SMX_REQUIRE(items[k].net && items[k].packed, SMX_E_NULL);
        const smx_mlp3_t& m = *items[k].net;
        SMX_REQUIRE(m.D > 0 && m.H1 > 0 && m.H2 > 0 && m.OUT > 0, SMX_E_SHAPE);
        SMX_REQUIRE(((uintptr_t)items[k].packed & 15) == 0, SMX_E_ALIGN);
        P.n[k].W1 = m.W1; P.n[k].W2 = m.W2; P.n[k].W3 = m.W3; P.n[k].packed = items[k].packed;
        P.n[k].D = m.D; P.n[k].H1 = m.H1; P.n[k].H2 = m.H2; P.n[k].OUT = m.OUT;
        P.n[k].base = base;
        base += smx_epoch_packed_floats(m.D, m.H1, m.H2, m.OUT) / 4;
    }
    P.total = base;
    return SMX_OK;
}

extern "C" int smx_epoch_pack_f32(const smx_epoch_pack_t* items, int32_t n, smx_stream_t stream) {
    PackArgs P;
    const int rc = fill_pack(P, items, n);
    if (rc) return rc;
    hipLaunchKernelGGL(epoch_pack_kernel, dim3((unsigned)((P.total + 255) / 256)), dim3(256), 0, smx_s(stream), P);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_epoch_prepare_f32(const smx_epoch_prep_t* a, smx_stream_t stream) {
    SMX_REQUIRE(a && a->obs0 && a->xn && a->xr, SMX_E_NULL);
    SMX_REQUIRE(a->rows > 0 && a->D > 0 && a->ld_obs0 >= a->D, SMX_E_SHAPE);
    SMX_REQUIRE((a->zmean == nullptr) == (a->zstd == nullptr), SMX_E_NULL);
    SMX_REQUIRE(!a->xnT || a->ldT >= a->rows, SMX_E_SHAPE);
    SMX_REQUIRE(!a->ref_filter || (a->ref_sum && a->ref_sumsq && a->ref_count), SMX_E_NULL);
    SMX_REQUIRE(!a->xnext || (a->obs_next && a->ld_next >= a->D), SMX_E_SHAPE);
    SMX_REQUIRE(!a->ref_std || (a->ref_log_var && a->A > 0 && a->ld_ref >= a->A), SMX_E_SHAPE);
    SMX_REQUIRE(a->n_zero >= 0 && a->n_zero <= (1 << 20) && (a->n_zero == 0 || a->zero_words), SMX_E_SHAPE);
    PrepArgs P;
    memset(&P, 0, sizeof(P));
    P.obs0 = a->obs0; P.ld_obs0 = (long)a->ld_obs0; P.rows = (long)a->rows; P.D = a->D;
    P.zmean = a->zmean; P.zstd = a->zstd; P.xn = a->xn; P.xnT = a->xnT; P.ldT = (long)a->ldT;
    P.rsum = a->ref_sum; P.rsumsq = a->ref_sumsq; P.rcount = a->ref_count; P.reps = a->ref_eps;
    P.ref_filter = a->ref_filter; P.xr = a->xr;
    P.obs_next = a->obs_next; P.ld_next = (long)a->ld_next; P.xnext = a->xnext;
    P.ref_log_var = a->ref_log_var; P.A = a->A; P.ref_std = a->ref_std; P.ld_ref = (long)a->ld_ref;
    P.zero_words = a->zero_words; P.n_zero = a->n_zero;
    P.n_x = P.rows * P.D;
    P.n_std = a->ref_std ? P.rows * a->A : 0;
    P.pack.count = 0; P.pack.total = 0;
    if (a->n_pack > 0) {
        const int rc = fill_pack(P.pack, a->pack, a->n_pack);
        if (rc) return rc;
    }
    const long total = 2 * P.n_x + (P.xnext ? P.n_x : 0) + P.n_std + P.pack.total;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(epoch_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, smx_s(stream), P);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// SMX_EPOCH_TIMING builds only (not declared in include/surreal_amd.h): where the timestamps go
extern "C" void smx_epoch_debug_tbuf(void* p) { g_tbuf = (long long*)p; }

extern "C" int32_t smx_epoch_blocks(int64_t rows) { return (int32_t)((rows + ER - 1) / ER); }

// LDS bytes of a forward launch whose widest job is (D, H1, H2) with A-dimensional policy losses: the same
// carve-up fill_args() makes ([x tile | h1 tile | h2 tile | out tile | K-split partials | loss scratch + inputs])
static int fwd_lds_bytes(int D, int H1, int H2, int A) {
    const int floats = ER * (r64(D) + 4) + ER * (r64(H1) + 4) + ER * (r64(H2) + 4) + ER * LDO + NWV * 2 * 256 +
                       loss_scratch_floats(A) + ER * (5 * A + 1);
    return floats * (int)sizeof(float);
}
// ... of a forward + backward launch: two dz3 tiles and the dz2 tile more, and the loss scratch doubles as the
// staging buffer of the partial-row reduction
static int fb_loss_floats(int A) {
    const int a = loss_scratch_floats(A) + ER * (5 * A + 1), b = FIN_CH * (8 + 2 * A);
    return a > b ? a : b;
}
static int fb_lds_bytes(int D, int H1, int H2, int A) {
    const int floats = ER * (r64(D) + 4) + ER * (r64(H1) + 4) + ER * (r64(H2) + 4) + ER * LDO + 2 * ER * LDZ +
                       ER * (r64(H2) + 4) + NWV * 2 * 256 + fb_loss_floats(A);
    return floats * (int)sizeof(float);
}
constexpr int MAX_LDS = 128 * 1024;

extern "C" int32_t smx_epoch_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    // (the LDS bound is taken for the widest loss a job of this network can carry: A = OUT; a launch pairs
    // networks that share D and the hidden sizes, so what this accepts fill_args() can place)
    return D > 0 && H1 > 0 && H2 > 0 && OUT > 0 && H1 % 4 == 0 && H2 % 4 == 0 && OUT <= 32 &&
           D <= 2048 && H1 <= 16 * NWV * TG * 2 && H2 <= 16 * NWV * TG * 2 &&
           fwd_lds_bytes(D, H1, H2, OUT) <= MAX_LDS;
}

extern "C" int32_t smx_epoch_fwdbwd_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    // (the dz2 accumulators of both right-hand sides live in registers across the wait: one pass of FTG tiles per wave)
    return smx_epoch_supported(D, H1, H2, OUT) && H2 <= 16 * FNWV * FTG && fb_lds_bytes(D, H1, H2, OUT) <= MAX_LDS;
}

enum { KIND_FWD = 0, KIND_BWD = 1, KIND_FB = 2 };

static int fill_args(EArgs& G, const smx_epoch_job_t* jobs, int32_t njobs, const smx_ppo_losses_t* loss,
                     int64_t n_total, int fsplit, int kind) {
    const bool backward = kind == KIND_BWD;
    SMX_REQUIRE(jobs, SMX_E_NULL);
    SMX_REQUIRE(njobs >= 1 && njobs <= MAX_EJOBS, SMX_E_SHAPE);
    memset(&G, 0, sizeof(G));
    G.n = njobs;
    G.n_total = n_total;
    G.fsplit = fsplit;
    G.tbuf = g_tbuf;
    int base = 0, maxD = 0, maxH1 = 0, maxH2 = 0, A = 0;
    for (int k = 0; k < njobs; ++k) {
        const smx_epoch_job_t& s = jobs[k];
        SMX_REQUIRE(s.net && s.x, SMX_E_NULL);
        const smx_mlp3_t& n = *s.net;
        SMX_REQUIRE(smx_epoch_supported(n.D, n.H1, n.H2, n.OUT), SMX_E_UNSUPPORTED);
        SMX_REQUIRE(s.rows > 0 && s.rows < (1 << 30), SMX_E_SHAPE);
        // 16-byte words are read from x rows (when D % 4 == 0), W2 / W3 rows and the hidden biases; W1
        // rows are D floats apart and their fragment loads only need 4-byte alignment
        SMX_REQUIRE(((n.D & 3) || ((uintptr_t)s.x & 15) == 0) && ((uintptr_t)n.W1 & 3) == 0 &&
                        ((uintptr_t)n.W2 & 15) == 0 && ((uintptr_t)n.W3 & 15) == 0 && ((uintptr_t)n.b1 & 15) == 0 &&
                        ((uintptr_t)n.b2 & 15) == 0,
                    SMX_E_ALIGN);
        SMX_REQUIRE((!s.h1T && !s.h2T) || (s.h1T && s.h2T && s.ldT >= s.rows), SMX_E_SHAPE);
        EJob& J = G.j[k];
        J.W1 = n.W1; J.b1 = n.b1; J.W2 = n.W2; J.b2 = n.b2; J.W3 = n.W3; J.b3 = n.b3;
        J.D = n.D; J.H1 = n.H1; J.H2 = n.H2; J.OUT = n.OUT;
        J.x = s.x; J.rows = (int)s.rows; J.h1T = s.h1T; J.h2T = s.h2T; J.ldT = (long)s.ldT;
        J.out = s.out; J.out_ld = s.out_ld ? s.out_ld : n.OUT; J.out_act = s.out_act; J.loss = s.loss;
        J.stop = s.stop_flag;
        J.dz3 = s.dz3; J.dz3T = s.dz3T; J.dz2T = s.dz2T; J.dz1T = s.dz1T;

        {
            SMX_REQUIRE(s.packed, SMX_E_NULL);
            SMX_REQUIRE(((uintptr_t)s.packed & 15) == 0, SMX_E_ALIGN);
            J.P1 = s.packed;
            J.P2 = s.packed + 4 * pack_off(n.D, n.H1, n.H2, n.OUT, 1);
            J.P3 = s.packed + 4 * pack_off(n.D, n.H1, n.H2, n.OUT, 2);
            J.P2T = s.packed + 4 * pack_off(n.D, n.H1, n.H2, n.OUT, 3);
            J.P3T = s.packed + 4 * pack_off(n.D, n.H1, n.H2, n.OUT, 4);
        }
        J.blk_base = base;
        base += smx_epoch_blocks(s.rows) * fsplit;
        if (s.loss == SMX_EPOCH_RHS_SURR || s.loss == SMX_EPOCH_RHS_KL) {
            SMX_REQUIRE(backward, SMX_E_UNSUPPORTED);
            SMX_REQUIRE(loss && loss->g_surr && loss->g_kl && s.dz3T, SMX_E_NULL);
        }
        if (s.loss == SMX_EPOCH_LOSS_POLICY || s.loss == SMX_EPOCH_RHS_SURR || s.loss == SMX_EPOCH_RHS_KL) {
            SMX_REQUIRE(loss, SMX_E_NULL);
            SMX_REQUIRE(loss->A == n.OUT && loss->A <= MAX_A, SMX_E_SHAPE);
            SMX_REQUIRE(loss->rows == s.rows, SMX_E_SHAPE);
            A = loss->A;
        }
        if (s.loss == SMX_EPOCH_LOSS_VALUE) {
            SMX_REQUIRE(loss && n.OUT == 1, SMX_E_SHAPE);
            SMX_REQUIRE(backward ? s.dz3 != nullptr : (loss->returns && loss->v_dz3 && loss->v_partials), SMX_E_NULL);
        }
        if (backward || kind == KIND_FB) SMX_REQUIRE(s.h1T && s.h2T && s.dz2T && s.dz1T, SMX_E_NULL);
        if (kind == KIND_FB) {
            SMX_REQUIRE(s.loss == SMX_EPOCH_LOSS_POLICY || s.loss == SMX_EPOCH_LOSS_VALUE, SMX_E_UNSUPPORTED);
            SMX_REQUIRE(smx_epoch_fwdbwd_supported(n.D, n.H1, n.H2, n.OUT), SMX_E_UNSUPPORTED);
            SMX_REQUIRE(s.loss != SMX_EPOCH_LOSS_POLICY || s.dz3T, SMX_E_NULL);
        }
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
    if (!backward && njobs == 2 && fsplit == 1) {         // two jobs of equal size on the two halves of the XCDs (xcd_job_order)
        const int nb0 = smx_epoch_blocks(jobs[0].rows), nb1 = smx_epoch_blocks(jobs[1].rows);
        static const bool off = getenv("SMX_EPOCH_NO_XSPLIT") != nullptr;     // A/B switch for measurements
        if (nb0 == nb1 && (nb0 & 3) == 0 && !off) G.xsplit = nb0;
    }
    // LDS carve-up: [x tile | h1 tile | h2 tile | out tile | K-split partials | loss scratch]
    // (the K loops run over an even number of 32-wide chunks: rows are zero padded to 64 columns)
    G.ldx = r64(maxD) + 4; G.ldh1 = r64(maxH1) + 4; G.ldh2 = r64(maxH2) + 4;
    if (backward) { G.ldx = 0; G.ldh1 = 0; }     // dz3 tile (at 0, stride LDO) | dz2 tile
    G.off_h1 = backward ? ER * LDZ : ER * G.ldx;
    G.off_h2 = G.off_h1 + ER * G.ldh1;
    G.off_out = G.off_h2 + ER * G.ldh2;
    // (forward + backward: the dz3 / dz2 tiles sit in front of off_red, inside the range the prologue clears)
    G.off_g = G.off_out + ER * LDO;
    G.off_red = G.off_out + (backward ? 0 : ER * LDO) + (kind == KIND_FB ? 2 * ER * LDZ + ER * G.ldh2 : 0);
    G.off_loss = G.off_red + (backward ? 0 : NWV * 2 * 256);
    const int loss_floats = backward ? FIN_CH * (8 + 2 * MAX_A)
                                     : (kind == KIND_FB ? fb_loss_floats(A) : loss_scratch_floats(A) + ER * (5 * A + 1));
    const int bytes = (G.off_loss + loss_floats) * (int)sizeof(float);
    // A launch has at most a few hundred workgroups, each keeping the four MFMA pipes of a CU busy
    // by itself: ask for more than half of the CU's 160 KB of LDS so that the dispatcher cannot put
    // two of them on one CU while others idle.
    return bytes > EXCLUSIVE_LDS ? bytes : EXCLUSIVE_LDS;
}

extern "C" int smx_epoch_forward_f32(const smx_epoch_job_t* jobs, int32_t njobs, const smx_ppo_losses_t* loss,
                                     smx_ppo_ctrl_t* ctrl, int64_t n_total, smx_stream_t stream) {
    EArgs G;
    const int lds = fill_args(G, jobs, njobs, loss, n_total, 1, KIND_FWD);
    if (lds < 0) return lds;
    SMX_REQUIRE(ctrl, SMX_E_NULL);
    if (loss && loss->mode != SMX_PPO_CLIP && loss->mode != SMX_PPO_ADAPT) return SMX_E_UNSUPPORTED;
    for (int k = 0; k < njobs; ++k)
        if (jobs[k].loss == SMX_EPOCH_LOSS_POLICY)
            SMX_REQUIRE(loss->log_var && loss->actions && loss->behave && loss->ref && loss->adv && loss->g_surr &&
                            loss->g_kl && loss->row_partials, SMX_E_NULL);
    const EJob& Lj = G.j[njobs - 1];
    const int blocks = Lj.blk_base + smx_epoch_blocks(Lj.rows);
    SMX_REQUIRE(lds <= MAX_LDS, SMX_E_UNSUPPORTED);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)epoch_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(epoch_fwd_kernel, dim3(blocks), dim3(FNTH), lds, smx_s(stream), G, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_epoch_backward_f32(const smx_epoch_job_t* jobs, int32_t njobs, const smx_ppo_losses_t* loss,
                                      smx_ppo_ctrl_t* ctrl, int64_t n_total, smx_stream_t stream);

extern "C" int smx_epoch_fwdbwd_f32(const smx_epoch_job_t* jobs, int32_t njobs, const smx_ppo_losses_t* loss,
                                    smx_ppo_ctrl_t* ctrl, int64_t n_total, int32_t* sync_word, uint64_t* kl_slots,
                                    smx_stream_t stream) {
    EArgs G;
    SMX_REQUIRE(loss && ctrl, SMX_E_NULL);
    const int lds = fill_args(G, jobs, njobs, loss, n_total, 1, KIND_FB);
    if (lds < 0) return lds;
    if (loss->mode != SMX_PPO_CLIP && loss->mode != SMX_PPO_ADAPT) return SMX_E_UNSUPPORTED;
    int n_policy = 0;
    for (int k = 0; k < njobs; ++k)
        if (jobs[k].loss == SMX_EPOCH_LOSS_POLICY) {
            // an updating epoch: the forward-only final pass stays on smx_epoch_forward_f32 + _backward_f32
            SMX_REQUIRE(loss->will_update, SMX_E_UNSUPPORTED);
            SMX_REQUIRE(loss->log_var && loss->actions && loss->behave && loss->ref && loss->adv && loss->g_surr &&
                            loss->g_kl && loss->row_partials && loss->dlogvar && loss->stats && sync_word && kl_slots,
                        SMX_E_NULL);
            SMX_REQUIRE(((uintptr_t)kl_slots & 7) == 0, SMX_E_ALIGN);
            ++n_policy;
        }
    SMX_REQUIRE(n_policy <= 1, SMX_E_SHAPE);       // one counter, one set of partial rows
    {
        // Adapt mode: every actor workgroup waits for the KL sums of ALL actor workgroups, so all of them must be
        // resident at once -- one per CU (the launch asks for more than half of a CU's LDS).  A launch with more
        // workgroups than CUs (more than 16 x CUs rows) would wait for workgroups that cannot start:
        // it runs as the two launches it would replace (same results, the batch means cross a launch boundary).
        // Clip mode never waits inside the launch except in the grid's LAST workgroup, which is dispatched after
        // every other one: safe at any size.
        const int n_cu = smx_cu_count();
        const EJob& Lj0 = G.j[njobs - 1];
        const int blocks0 = Lj0.blk_base + smx_epoch_blocks(Lj0.rows);
        if (n_policy && loss->mode == SMX_PPO_ADAPT && blocks0 > n_cu) {
            const int rc = smx_epoch_forward_f32(jobs, njobs, loss, ctrl, n_total, stream);
            if (rc) return rc;
            return smx_epoch_backward_f32(jobs, njobs, loss, ctrl, n_total, stream);
        }
    }
    G.sync = (int*)sync_word;
    G.kl_slots = (unsigned long long*)kl_slots;
    for (int k = 0; k < njobs; ++k)
        if (jobs[k].loss == SMX_EPOCH_LOSS_POLICY) {
            G.nb_pol = smx_epoch_blocks(jobs[k].rows);
            G.pol_stop = jobs[k].stop_flag;
        }
    const EJob& Lj = G.j[njobs - 1];
    const int blocks = Lj.blk_base + smx_epoch_blocks(Lj.rows);
    SMX_REQUIRE(lds <= MAX_LDS, SMX_E_UNSUPPORTED);
    static bool attr_set_fb = false;
    if (!attr_set_fb) {
        (void)hipFuncSetAttribute((const void*)epoch_fb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set_fb = true;
    }
    hipLaunchKernelGGL(epoch_fb_kernel, dim3(blocks), dim3(FNTH), lds, smx_s(stream), G, ctrl);
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
    const int lds = fill_args(G, jobs, njobs, loss, n_total, fs, KIND_BWD);
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
    static bool attr_set_b = false;
    if (!attr_set_b) {
        (void)hipFuncSetAttribute((const void*)epoch_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set_b = true;
    }
    hipLaunchKernelGGL(epoch_bwd_kernel, dim3(blocks), dim3(NTH), lds, smx_s(stream), G, ctrl);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

// ---------------------------------------------------------------------------------------------
// smx_device_occupy: `blocks` workgroups that each hold a compute unit to themselves (more than half of its LDS) for
// `microseconds`, doing nothing.  What a tenant of a shared device looks like to the launches above: the tests put it
// on a second stream under the fused forward + backward epoch (tests/test_gpu_epoch.py) to show that the in-launch
// wait rides out a co-resident kernel and that the learner falls back when it cannot.
// ---------------------------------------------------------------------------------------------
__global__ void occupy_kernel(long long ticks, int* sink) {
    extern __shared__ float occ[];
    const long long t0 = (long long)wall_clock64();
    float acc = 0.f;
    while ((long long)wall_clock64() - t0 < ticks) {
        occ[threadIdx.x] = acc;
        acc += occ[(threadIdx.x + 1) & 63];
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 12345.678f && sink) *sink = 1;      // (keeps the LDS traffic alive)
}

extern "C" int smx_device_occupy(int32_t blocks, int64_t microseconds, smx_stream_t stream) {
    SMX_REQUIRE(blocks > 0 && blocks <= 4096 && microseconds >= 0 && microseconds <= 2000000, SMX_E_SHAPE);
    static bool attr_set_o = false;
    if (!attr_set_o) {
        (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set_o = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), EXCLUSIVE_LDS, smx_s(stream), (long long)microseconds * 100LL,
                       (int*)nullptr);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
