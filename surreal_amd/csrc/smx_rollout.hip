// A whole device-resident rollout in ONE launch (surreal/agent/base.py:244-271 the per-step loop of a rollout
// worker, surreal/agent/ppo_agent.py:106-154 act: z-filter -> policy MLP -> DiagGauss sample -> clip; the environment
// step and the recording that surreal/env/exp_sender_wrapper.py:153-264 does on the host).
//
// Actors are independent of each other, time steps are not: so a workgroup OWNS 16 actors and walks them through all
// T steps by itself -- no per-step launch, no grid-wide synchronisation.  Per step:
//   x tile (z-filtered observations of its 16 actors, LDS)  ->  the three policy layers on v_mfma_f32_16x16x4_f32 with
//   the fragment-order packed weights streamed from L2 (the row-block loop of smx_epoch_mma.inc.h, the same operations
//   in the same order as smx_epoch_forward_f32: bit-identical means)  ->  the sampling head (mean, std * exp(noise),
//   a = clip(mean + std * eps))  ->  the synthetic environment's step for the 16 actors (state kept in registers for
//   the whole rollout)  ->  the transition recorded into the rollout tables [actors, T + 1, .]  ->  the next observation
//   z-filtered straight into the x tile.
// The per-step launches it replaces were 3 dependent launches of ~9.5 us each (two hidden layers as GEMM launches, then
// head + step), 384 launches for T = 128; here a step is the MFMA issue time of one CU for 16 rows (~12 us at
// D = 376, [300, 200]) and the chip runs 256 such workgroups side by side (4096 actors cost what 1024 do).
#include "smx_common.h"
#include <stdlib.h>
#include <string.h>

namespace {
#include "smx_epoch_pack.inc.h"
#include "smx_epoch_mma.inc.h"
#include "smx_rows4_mma.inc.h"

// Phase timestamps exist only in a build with -DSMX_ROLLOUT_TIMING (scripts/bench_rollout.py); the product build has none.
#ifdef SMX_ROLLOUT_TIMING
#define RSTAMP(i) do { if (g_rtbuf_dev && threadIdx.x == 0 && step == G.steps / 2) g_rtbuf_dev[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#define RWALL(i) do { if (g_rtbuf_dev && threadIdx.x == 0) g_rtbuf_dev[(size_t)blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)
#define RCYC(i) do { if (g_rtbuf_dev && threadIdx.x == 0) g_rtbuf_dev[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
__device__ long long* g_rtbuf_dev = nullptr;
#else
#define RSTAMP(i) do { } while (0)
#define RWALL(i) do { } while (0)
#define RCYC(i) do { } while (0)
#endif

constexpr int RLDO = 36;          // row stride of the mean tile in LDS (<= 32 actions)
constexpr int RMAX_A = 32;

struct RollArgs {
    const float *P1, *P2, *P3;                  // packed weights (smx_epoch_pack_f32)
    const float *b1, *b2, *b3;
    int D, H1, H2, A, out_act;
    const float *log_var, *noise_scale, *eps;   // eps [T, n, A] or null (deterministic)
    const float *zsum, *zsumsq, *zcount;        // z-filter running sums or null
    float zeps;
    float* state;                               // [n, D] in / out
    const float* init_state;
    int n, t0, episode_len, steps, R, slot0;    // R = rows per actor in the rollout tables
    float *obs_roll, *act_roll, *rew_roll, *done_roll, *pd_roll, *obs_last;
    int ldx, ldh1, ldh2, off_h1, off_h2, off_out, off_act, off_red3, off_z, lds_floats;
};

__device__ __forceinline__ float zclamp_r(float x, float m, float sd) {
    float v = (x - m) / sd;                     // z_filter.py:77
    if (v == v) v = fminf(fmaxf(v, -5.0f), 5.0f);
    return v;
}

constexpr int RNWV = 8;           // two wavefronts per SIMD: the K loops have no barrier inside, so one wave's loads
constexpr int RNTH = 64 * RNWV;   // and epilogue hide under the other's MFMAs
constexpr int RKV = 8;            // observation elements a lane owns per row (D <= 64 RKV)

// RG row groups of four actors per workgroup (round 6; smx_rows4_mma.inc.h).  1024 actors: RG = 1 -> 256 workgroups, one per
// CU.  The host picks the smallest RG whose grid fits the chip once (more actors per workgroup = fewer passes over the
// packed weights per actor; fewer = more CUs at work).
template <int RG>
__global__ __launch_bounds__(RNTH) void rollout_kernel(RollArgs G) {
    constexpr int RB = 4 * RG;                       // actors per workgroup
    constexpr int WPR = RB < RNWV ? RNWV / RB : 1;   // wavefronts per actor row in the environment phase
    constexpr int RPW = RB > RNWV ? RB / RNWV : 1;   // actor rows per wavefront
    constexpr int KPL = RKV / WPR;                   // observation elements a lane owns per row
    constexpr int NT = 3;                            // feature tiles a wave carries per pass
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 15, kq = lane >> 4;
    const long row0 = (long)blockIdx.x * RB;
    int nrows = G.n - (int)row0;
    if (nrows > RB) nrows = RB;
    const int D = G.D, A = G.A, R = G.R;
    float* xs = sm;
    float* h1s = sm + G.off_h1;
    float* h2s = sm + G.off_h2;
    float* outs = sm + G.off_out;
    float* s_act = sm + G.off_act;              // [RB][RMAX_A] clipped actions
    const int ldx = G.ldx, ldh1 = G.ldh1, ldh2 = G.ldh2;

    // ---- once: clear the tiles; the z-filter's mean / std and k % A (an integer division per element and step
    // otherwise) go to LDS tables; the environment phase's lanes keep the raw state of their elements in registers for
    // the whole rollout: wave wv owns rows RPW wv .. (RB >= 8) or the part `part` of row wv / WPR (RB = 4), a lane the
    // elements k = lane + 64 (part + WPR i)
    for (int i = tid; i < G.off_z; i += RNTH) sm[i] = 0.f;          // (incl. the action tile: its unused columns stay 0)
    float* zm = sm + G.off_z;                   // [D] z-filter mean | [D] std | [D] k % A (as int)
    float* zs = zm + D;
    int* kmod = (int*)(zs + D);
    for (int k = tid; k < D; k += RNTH) {
        kmod[k] = k % A;
        float m = 0.f, sz = 1.f;
        if (G.zsum) {
            const float c = G.zcount[0];
            m = G.zsum[k] / c;
            const float var = G.zsumsq[k] / c - m * m;
            sz = sqrtf(var);
            if (sz == sz) sz = fmaxf(sz, G.zeps);
        }
        zm[k] = m; zs[k] = sz;
    }
    const int erow0 = RPW * (wv / WPR), part = wv % WPR;
    float st[RPW][KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int k = lane + 64 * (part + WPR * i);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = erow0 + rr;
            st[rr][i] = (k < D && r < nrows) ? G.state[(row0 + r) * D + k] : 0.f;
        }
    }
    SMX_LDS_BARRIER();
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = erow0 + rr;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int k = lane + 64 * (part + WPR * i);
            if (k < D && r < nrows) xs[r * ldx + k] = G.zsum ? zclamp_r(st[rr][i], zm[k], zs[k]) : st[rr][i];
        }
    }
    SMX_LDS_BARRIER();

    // ---- the output layer (K = H2, <= 32 outputs) is a latency chain if two waves walk its chunks alone (measured: 5.0 k
    // cycles of a 36 k-cycle step for 2 tiles x 8 chunks).  Its K is split over the EIGHT waves instead: wave w owns chunk w
    // of both tiles, keeps those 4 KB of packed weights in registers for the whole rollout (no loads at all), and the
    // eight partial sums of a (row, action) pair meet in the sampling head, added in wave order.  Shapes with more than 8
    // chunks (H2 > 256) or more than 2 tiles take the generic loop.
    const int tiles3 = (A + 15) >> 4, C3 = pack_chunks(G.H2);
    const bool l3res = C3 <= RNWV && tiles3 <= 2;
    float4 w3a[2], w3b[2];
    {
        const rsrc_t rw3 = make_rsrc(G.P3, (unsigned)tiles3 * (unsigned)C3 * 2048u);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const unsigned o = (l3res && g < tiles3 && wv < C3)
                ? (((unsigned)g * (unsigned)C3 + (unsigned)wv) * 512u + (unsigned)lane * 4u) * 4u : OOB;
            w3a[g] = ld16(rw3, o);
            w3b[g] = ld16(rw3, o == OOB ? OOB : o + 1024u);
        }
    }
    float* red3 = sm + G.off_red3;               // [RNWV][RB][32]: the waves' partial output sums
    // the sampling head's per-pair constants (the same expressions, formed once instead of every step)
    const int hr = tid / A, hj = tid - hr * A;            // RB x A <= 512 pairs
    const float b3v = G.b3[hj];
    float sd0 = expf(G.log_var[hj]);
    if (G.noise_scale && hr < nrows) sd0 = sd0 * G.noise_scale[row0 + hr];
    // the environment phase's per-element constants: the action column k % A and the drift term (an LDS round trip in front
    // of the action read and an integer modulo per element and step otherwise)
    int am[KPL];
    float dr[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int k = lane + 64 * (part + WPR * i);
        am[i] = k < D ? kmod[k] : 0;
        dr[i] = 0.01f * (float)(((37 * k) % 17) - 8);
    }
    int t = G.t0;
    RWALL(12); RCYC(13);
#pragma unroll 1
    for (int step = 0; step < G.steps; ++step) {
        const int slot = G.slot0 + step;
        const bool last_step = step + 1 == G.steps;
        RSTAMP(0);
        // this step's normal draw of the lane's (row, action) pair, requested before the layers (consumed behind them)
        float ev = 0.f;
        if (G.eps && hr < nrows) ev = G.eps[((size_t)step * G.n + row0 + hr) * A + hj];
        // ---- the three layers ------------------------------------------------------------------------------
#pragma unroll 1
        for (int l = 0; l < (l3res ? 2 : 3); ++l) {
            const float* Wp = l == 0 ? G.P1 : (l == 1 ? G.P2 : G.P3);
            const float* bias = l == 0 ? G.b1 : (l == 1 ? G.b2 : G.b3);
            const int H = l == 0 ? G.H1 : (l == 1 ? G.H2 : A);
            const int K = l == 0 ? D : (l == 1 ? G.H1 : G.H2);
            const float* in_lds = l == 0 ? xs : (l == 1 ? h1s : h2s);
            const int ldi = l == 0 ? ldx : (l == 1 ? ldh1 : ldh2);
            float* out_lds = l == 0 ? h1s : (l == 1 ? h2s : outs);
            const int ldo = l == 0 ? ldh1 : (l == 1 ? ldh2 : RLDO);
            const int tiles = (H + 15) >> 4;
            const int C2 = pack_chunks(K);
            const rsrc_t rw = make_rsrc(Wp, (unsigned)tiles * (unsigned)C2 * 2048u);
            const rsrc_t rbias = make_rsrc(bias, (unsigned)H * 4u);
#pragma unroll 1
            for (int tb = 0; tb < tiles; tb += RNWV * NT) {
                const int t0 = tb + wv;
                if (t0 >= tiles) continue;                            // (wave-uniform)
                float bs[NT];
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    const int f = 16 * (t0 + RNWV * g) + fm;
                    bs[g] = ld4(rbias, (f < H) ? (unsigned)f * 4u : OOB);
                }
                f32x4 acc[NT][RG];
#pragma unroll
                for (int g = 0; g < NT; ++g)
#pragma unroll
                    for (int r = 0; r < RG; ++r) acc[g][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                fwd_tiles4<NT, RG, true>(acc, rw, tiles, C2, in_lds, ldi, t0, RNWV, lane);
                // the kq groups meet; lane (fm, kq) then keeps row kq of every group: bias, activation, one word to LDS
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    const int f = 16 * (t0 + RNWV * g) + fm;
                    if (t0 + RNWV * g < tiles) {                      // (wave-uniform)
#pragma unroll
                        for (int r = 0; r < RG; ++r) {
                            float z = meet_rows(acc[g][r]);
                            z += bs[g];
                            if (l == 2) z = act_f(z, G.out_act);
                            else z = (z < 0.f) ? 0.f : z;
                            out_lds[(4 * r + kq) * ldo + f] = (f < H) ? z : 0.f;
                        }
                    }
                }
            }
            SMX_LDS_BARRIER();
            RSTAMP(1 + l);
        }
        if (l3res) {
            // wave w: chunk w of the output layer against h2 (zero weights past the last chunk: a zero partial sum)
            const float* bp = h2s + (lane & 3) * ldh2 + 8 * kq + 32 * (wv < C3 ? wv : 0);
            f32x4 a3[2][RG];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < RG; ++r) a3[g][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float4 x0[RG], x1[RG];
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                x0[r] = *(const float4*)(bp + 4 * r * ldh2);
                x1[r] = *(const float4*)(bp + 4 * r * ldh2 + 4);
            }
#define SMX_L3(X, W, E)                                                      \
            _Pragma("unroll") for (int g = 0; g < 2; ++g)                    \
                _Pragma("unroll") for (int r = 0; r < RG; ++r) a3[g][r] = MFMA4(X[r].E, W[g].E, a3[g][r]);
            SMX_L3(x0, w3a, x) SMX_L3(x0, w3a, y) SMX_L3(x0, w3a, z) SMX_L3(x0, w3a, w)
            SMX_L3(x1, w3b, x) SMX_L3(x1, w3b, y) SMX_L3(x1, w3b, z) SMX_L3(x1, w3b, w)
#undef SMX_L3
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const float z = meet_rows(a3[g][r]);
                    red3[(wv * RB + 4 * r + kq) * 32 + 16 * g + fm] = z;
                }
            SMX_LDS_BARRIER();
            RSTAMP(3);
        }
        // ---- sampling head (smx_diaggauss_sample_f32's expressions): one (actor, action) pair per lane -------
        const bool done = (t + 1 >= G.episode_len);
        if (hr < nrows) {
            const long a = row0 + hr;
            float mu;
            if (l3res) {
                float p8[RNWV];
#pragma unroll
                for (int w = 0; w < RNWV; ++w) p8[w] = red3[(w * RB + hr) * 32 + hj];
                float z = p8[0];
#pragma unroll
                for (int w = 1; w < RNWV; ++w) z += p8[w];
                mu = act_f(z + b3v, G.out_act);
            } else {
                mu = outs[hr * RLDO + hj];
            }
            const float sd = sd0;
            float act = G.eps ? ev * sd + mu : mu;
            if (act == act) act = fminf(fmaxf(act, -1.0f), 1.0f);
            s_act[hr * RMAX_A + hj] = act;
            // (the rollout tables are written once and read by a later launch: streaming stores, so that 3 MB of them per
            // step do not push the packed weights -- re-read by every workgroup every step -- out of the L2s)
            if (G.act_roll) __builtin_nontemporal_store(act, &G.act_roll[(a * R + slot) * A + hj]);
            if (G.pd_roll) {
                __builtin_nontemporal_store(mu, &G.pd_roll[(a * R + slot) * 2 * A + hj]);
                __builtin_nontemporal_store(sd, &G.pd_roll[(a * R + slot) * 2 * A + A + hj]);
            }
        }
        SMX_LDS_BARRIER();
        RSTAMP(4);
        // ---- environment step of the actors (smx_synth_env_step_f32's expressions), recording, next x tile ------
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = erow0 + rr;                      // wave-uniform
            if (r < nrows) {
                const long a = row0 + r;
                float* orow = G.obs_roll ? G.obs_roll + (a * R + slot) * D : nullptr;
                float sn0 = 0.f;
#pragma unroll
                for (int i = 0; i < KPL; ++i) {
                    const int k = lane + 64 * (part + WPR * i);
                    if (k < D) {
                        const float ac = s_act[r * RMAX_A + am[i]];
                        const float s = st[rr][i];
                        const float drift = dr[i];
                        float sn = (0.9f * s + 0.5f * ac) + drift;
                        sn = fminf(fmaxf(sn, -10.0f), 10.0f);
                        if (orow) {
                            __builtin_nontemporal_store(s, &orow[k]);
                            // the observation AFTER the step: row slot + 1 -- which the next step of this launch writes
                            // itself (the same value, or the reset state when the episode ended here), so only the
                            // launch's last step stores it
                            if (slot + 1 < R) { if (last_step) __builtin_nontemporal_store(sn, &orow[D + k]); }
                            else if (G.obs_last) G.obs_last[a * D + k] = sn;     // (the replay's obs_next field)
                        }
                        if (i == 0) sn0 = sn;
                        const float next = done ? G.init_state[a * D + k] : sn;
                        st[rr][i] = next;
                        xs[r * ldx + k] = G.zsum ? zclamp_r(next, zm[k], zs[k]) : next;
                    }
                }
                if (lane == 0 && part == 0) {               // (k == 0 lives in lane 0, i == 0 of the row's first wave)
                    // sum_j a_j^2 in fp64, j ascending (the order of smx_synth_env_step_f32).  All RMAX_A reads are
                    // issued up front (unused columns of the tile are zero and add +0.0): one LDS round trip, not A
                    float av[RMAX_A];
#pragma unroll
                    for (int j = 0; j < RMAX_A; ++j) av[j] = s_act[r * RMAX_A + j];
                    double q = 0.0;
#pragma unroll
                    for (int j = 0; j < RMAX_A; ++j) q += (double)av[j] * (double)av[j];
                    if (G.rew_roll) __builtin_nontemporal_store((float)(-0.1 * q + 0.05 * (double)sn0), &G.rew_roll[a * R + slot]);
                    if (G.done_roll) __builtin_nontemporal_store(done ? 1.0f : 0.0f, &G.done_roll[a * R + slot]);
                }
            }
        }
        t = done ? 0 : t + 1;
        SMX_LDS_BARRIER();
        RSTAMP(5);
    }
    RWALL(14); RCYC(15);
    // ---- the states the actors are left in ---------------------------------------------------------------
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = erow0 + rr;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int k = lane + 64 * (part + WPR * i);
            if (k < D && r < nrows) G.state[(row0 + r) * D + k] = st[rr][i];
        }
    }
}

constexpr int RROWS = ER / RNWV;  // actor rows a wavefront steps in the environment phase
constexpr int RTG = 3;            // feature tiles a wave carries per pass (register budget of two waves per SIMD)

// More actors than 8 per CU (n > 2048 on 256 CUs): 16 actors per workgroup on v_mfma_f32_16x16x4_f32 (the row-block loop of
// smx_epoch_mma.inc.h, bit-identical means to smx_epoch_forward_f32) -- at 16 rows the matrix pipes bound the step and one
// 16x16x4 operand read feeds 16 rows where 4x4x1 needs four (measured, 4096 actors x 128 steps: 2.95 ms against 3.9 ms on
// four row groups of the 4-row loop).
__global__ __launch_bounds__(RNTH) void rollout16_kernel(RollArgs G) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 15, kq = lane >> 4;
    const long row0 = (long)blockIdx.x * ER;
    int nrows = G.n - (int)row0;
    if (nrows > ER) nrows = ER;
    const int D = G.D, A = G.A, R = G.R;
    float* xs = sm;
    float* h1s = sm + G.off_h1;
    float* h2s = sm + G.off_h2;
    float* outs = sm + G.off_out;
    float* s_act = sm + G.off_act;              // [16][RMAX_A] clipped actions
    const int ldx = G.ldx, ldh1 = G.ldh1, ldh2 = G.ldh2;

    // ---- once: clear the tiles; the z-filter's mean / std and k % A (an integer division per element and step
    // otherwise) go to LDS tables; a lane owns elements k = lane + 64 i of the actor rows 2 wv and 2 wv + 1 and keeps
    // their raw state in registers for the whole rollout
    for (int i = tid; i < G.off_z; i += RNTH) sm[i] = 0.f;          // (incl. the action tile: its unused columns stay 0)
    float* zm = sm + G.off_z;                   // [D] z-filter mean | [D] std | [D] k % A (as int)
    float* zs = zm + D;
    int* kmod = (int*)(zs + D);
    for (int k = tid; k < D; k += RNTH) {
        kmod[k] = k % A;
        float m = 0.f, sz = 1.f;
        if (G.zsum) {
            const float c = G.zcount[0];
            m = G.zsum[k] / c;
            const float var = G.zsumsq[k] / c - m * m;
            sz = sqrtf(var);
            if (sz == sz) sz = fmaxf(sz, G.zeps);
        }
        zm[k] = m; zs[k] = sz;
    }
    float st[RROWS][RKV];
#pragma unroll
    for (int i = 0; i < RKV; ++i) {
        const int k = lane + 64 * i;
#pragma unroll
        for (int rr = 0; rr < RROWS; ++rr) {
            const int r = RROWS * wv + rr;
            st[rr][i] = (k < D && r < nrows) ? G.state[(row0 + r) * D + k] : 0.f;
        }
    }
    SMX_LDS_BARRIER();
#pragma unroll
    for (int rr = 0; rr < RROWS; ++rr) {
        const int r = RROWS * wv + rr;
#pragma unroll
        for (int i = 0; i < RKV; ++i) {
            const int k = lane + 64 * i;
            if (k < D && r < nrows) xs[r * ldx + k] = G.zsum ? zclamp_r(st[rr][i], zm[k], zs[k]) : st[rr][i];
        }
    }
    SMX_LDS_BARRIER();

    int t = G.t0;
    RWALL(12); RCYC(13);
#pragma unroll 1
    for (int step = 0; step < G.steps; ++step) {
        const int slot = G.slot0 + step;
        RSTAMP(0);
        // this step's normal draw of the lane's (row, action) pair, requested before the layers (consumed behind them)
        const int hr = tid / A, hj = tid - hr * A;        // 16 x A <= 512 pairs
        float ev = 0.f;
        if (G.eps && hr < nrows) ev = G.eps[((size_t)step * G.n + row0 + hr) * A + hj];
        // ---- the three layers: the loop of epoch_fwd_kernel without its global stores ---------------------
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            const float* Wp = l == 0 ? G.P1 : (l == 1 ? G.P2 : G.P3);
            const float* bias = l == 0 ? G.b1 : (l == 1 ? G.b2 : G.b3);
            const int H = l == 0 ? G.H1 : (l == 1 ? G.H2 : A);
            const int K = l == 0 ? D : (l == 1 ? G.H1 : G.H2);
            const float* in_lds = l == 0 ? xs : (l == 1 ? h1s : h2s);
            const int ldi = l == 0 ? ldx : (l == 1 ? ldh1 : ldh2);
            float* out_lds = l == 0 ? h1s : (l == 1 ? h2s : outs);
            const int ldo = l == 0 ? ldh1 : (l == 1 ? ldh2 : RLDO);
            const int tiles = (H + 15) >> 4;
            const int C2 = pack_chunks(K);
            const rsrc_t rw = make_rsrc(Wp, (unsigned)tiles * (unsigned)C2 * 2048u);
            const rsrc_t rbias = make_rsrc(bias, (unsigned)H * 4u);
#pragma unroll 1
            for (int tb = 0; tb < tiles; tb += RNWV * RTG) {
                const int t0 = tb + wv;
                int nt = (tiles - t0 + RNWV - 1) / RNWV;
                nt = nt < 0 ? 0 : (nt > RTG ? RTG : nt);
                float bs[RTG][4];
#pragma unroll
                for (int g = 0; g < RTG; ++g) {
                    const int f0 = 16 * (t0 + RNWV * g) + 4 * kq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bs[g][r] = ld4(rbias, (g < nt) ? (unsigned)(f0 + r) * 4u : OOB);
                }
                f32x4 acc[TG];
#pragma unroll
                for (int g = 0; g < TG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (nt > 2) fwd_tiles<3>(acc, rw, tiles, C2, in_lds, ldi, t0, RNWV, lane);
                else if (nt > 1) fwd_tiles<2>(acc, rw, tiles, C2, in_lds, ldi, t0, RNWV, lane);
                else if (nt > 0) fwd_tiles<1>(acc, rw, tiles, C2, in_lds, ldi, t0, RNWV, lane);
#pragma unroll
                for (int g = 0; g < RTG; ++g) {
                    if (g < nt) {
                        const int f0 = 16 * (t0 + RNWV * g) + 4 * kq;
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float z = acc[g][r] + bs[g][r];
                            if (l == 2) z = act_f(z, G.out_act);
                            else z = (z < 0.f) ? 0.f : z;
                            v[r] = (f0 + r < H) ? z : 0.f;
                        }
                        *(float4*)(out_lds + fm * ldo + f0) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
            SMX_LDS_BARRIER();
            RSTAMP(1 + l);
        }
        // ---- sampling head (smx_diaggauss_sample_f32's expressions): one (actor, action) pair per lane -------
        const bool done = (t + 1 >= G.episode_len);
        if (hr < nrows) {
            const long a = row0 + hr;
            const float mu = outs[hr * RLDO + hj];
            float sd = expf(G.log_var[hj]);
            if (G.noise_scale) sd = sd * G.noise_scale[a];
            float act = G.eps ? ev * sd + mu : mu;
            if (act == act) act = fminf(fmaxf(act, -1.0f), 1.0f);
            s_act[hr * RMAX_A + hj] = act;
            if (G.act_roll) G.act_roll[(a * R + slot) * A + hj] = act;
            if (G.pd_roll) {
                G.pd_roll[(a * R + slot) * 2 * A + hj] = mu;
                G.pd_roll[(a * R + slot) * 2 * A + A + hj] = sd;
            }
        }
        SMX_LDS_BARRIER();
        RSTAMP(4);
        // ---- environment step of the 16 actors (smx_synth_env_step_f32's expressions), recording, next x tile ---
#pragma unroll
        for (int rr = 0; rr < RROWS; ++rr) {
            const int r = RROWS * wv + rr;                 // wave-uniform
            if (r < nrows) {
                const long a = row0 + r;
                float* orow = G.obs_roll ? G.obs_roll + (a * R + slot) * D : nullptr;
                float sn0 = 0.f;
#pragma unroll
                for (int i = 0; i < RKV; ++i) {
                    const int k = lane + 64 * i;
                    if (k < D) {
                        const float ac = s_act[r * RMAX_A + kmod[k]];
                        const float s = st[rr][i];
                        const float drift = 0.01f * (float)(((37 * k) % 17) - 8);
                        float sn = (0.9f * s + 0.5f * ac) + drift;
                        sn = fminf(fmaxf(sn, -10.0f), 10.0f);
                        if (orow) {
                            orow[k] = s;
                            if (slot + 1 < R) orow[D + k] = sn;
                            else if (G.obs_last) G.obs_last[a * D + k] = sn;     // (the replay's obs_next field)
                        }
                        if (i == 0) sn0 = sn;
                        const float next = done ? G.init_state[a * D + k] : sn;
                        st[rr][i] = next;
                        xs[r * ldx + k] = G.zsum ? zclamp_r(next, zm[k], zs[k]) : next;
                    }
                }
                if (lane == 0) {                            // (k == 0 lives in lane 0, i == 0)
                    // sum_j a_j^2 in fp64, j ascending (the order of smx_synth_env_step_f32).  All RMAX_A reads are
                    // issued up front (unused columns of the tile are zero and add +0.0): one LDS round trip, not A
                    float av[RMAX_A];
#pragma unroll
                    for (int j = 0; j < RMAX_A; ++j) av[j] = s_act[r * RMAX_A + j];
                    double q = 0.0;
#pragma unroll
                    for (int j = 0; j < RMAX_A; ++j) q += (double)av[j] * (double)av[j];
                    if (G.rew_roll) G.rew_roll[a * R + slot] = (float)(-0.1 * q + 0.05 * (double)sn0);
                    if (G.done_roll) G.done_roll[a * R + slot] = done ? 1.0f : 0.0f;
                }
            }
        }
        t = done ? 0 : t + 1;
        SMX_LDS_BARRIER();
        RSTAMP(5);
    }
    RWALL(14); RCYC(15);
    // ---- the states the actors are left in ---------------------------------------------------------------
#pragma unroll
    for (int rr = 0; rr < RROWS; ++rr) {
        const int r = RROWS * wv + rr;
#pragma unroll
        for (int i = 0; i < RKV; ++i) {
            const int k = lane + 64 * i;
            if (k < D && r < nrows) G.state[(row0 + r) * D + k] = st[rr][i];
        }
    }
}

inline int rr64(int v) { return (v + 63) & ~63; }

// row strides = 16 mod 64 words: the rows of a group (a word's lanes: row l & 3, k offset 8 (l >> 4)) and the epilogue's
// one-word stores (row kq, feature fm) fall on distinct banks of the 64.  A tile holds pack_chunks(K) * 32 + 8 columns at
// least (the loop's last prefetch reads one chunk it does not use).
// (16 rows on the 16x16x4 loop: a word's lanes are row l & 15 -- strides of 4 mod 64 spread them.)
int carve(RollArgs& G, int RB) {
    if (RB == 16) { G.ldx = rr64(G.D) + 4; G.ldh1 = rr64(G.H1) + 4; G.ldh2 = rr64(G.H2) + 4; }
    else { G.ldx = rr64(G.D + 40) + 16; G.ldh1 = rr64(G.H1 + 40) + 16; G.ldh2 = rr64(G.H2 + 40) + 16; }
    G.off_h1 = RB * G.ldx;
    G.off_h2 = G.off_h1 + RB * G.ldh1;
    G.off_out = G.off_h2 + RB * G.ldh2;
    G.off_act = G.off_out + RB * RLDO;
    G.off_red3 = G.off_act + RB * RMAX_A;
    G.off_z = G.off_red3 + RNWV * RB * 32;
    G.lds_floats = G.off_z + 3 * G.D;
    return G.lds_floats * (int)sizeof(float);
}

int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}

constexpr int ROLL_MAX_LDS = 150 * 1024;
constexpr int ROLL_EXCLUSIVE_LDS = 84 * 1024;

}  // namespace

#ifdef SMX_ROLLOUT_TIMING
// timing builds only (not declared in include/surreal_amd.h): where the timestamps go ([workgroups][16] int64)
extern "C" void smx_rollout_debug_tbuf(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rtbuf_dev), &p, sizeof(p)); }
#endif

extern "C" int32_t smx_synth_rollout_supported(int32_t D, int32_t H1, int32_t H2, int32_t A) {
    if (!(D > 0 && H1 > 0 && H2 > 0 && A > 0 && A <= RMAX_A && H1 % 4 == 0 && H2 % 4 == 0)) return 0;
    if (!(H1 <= 640 && H2 <= 640 && D <= 64 * RKV)) return 0;
    RollArgs G;
    memset(&G, 0, sizeof(G));
    G.D = D; G.H1 = H1; G.H2 = H2; G.A = A;
    return carve(G, 16) <= ROLL_MAX_LDS;
}

extern "C" int smx_synth_rollout_f32(const smx_synth_rollout_t* a, smx_stream_t stream) {
    SMX_REQUIRE(a && a->net && a->packed && a->log_var && a->state && a->init_state, SMX_E_NULL);
    const smx_mlp3_t& n = *a->net;
    SMX_REQUIRE(smx_synth_rollout_supported(n.D, n.H1, n.H2, n.OUT), SMX_E_UNSUPPORTED);
    SMX_REQUIRE(a->n > 0 && a->steps > 0 && a->episode_len > 0 && a->rows_per_actor > 0, SMX_E_SHAPE);
    // (every roll table is indexed by slot + step: the bound holds whichever of them is recorded)
    const bool records = a->obs_roll || a->act_roll || a->rew_roll || a->done_roll || a->pd_roll;
    SMX_REQUIRE(a->slot >= 0 && (!records || a->slot + a->steps <= a->rows_per_actor), SMX_E_SHAPE);
    SMX_REQUIRE(((uintptr_t)a->packed & 15) == 0 && ((uintptr_t)n.b1 & 3) == 0, SMX_E_ALIGN);
    SMX_REQUIRE((a->zsum == nullptr) == (a->zsumsq == nullptr) && (a->zsum == nullptr) == (a->zcount == nullptr), SMX_E_NULL);
    RollArgs G;
    memset(&G, 0, sizeof(G));
    G.P1 = a->packed;
    G.P2 = a->packed + 4 * pack_off(n.D, n.H1, n.H2, n.OUT, 1);
    G.P3 = a->packed + 4 * pack_off(n.D, n.H1, n.H2, n.OUT, 2);
    G.b1 = n.b1; G.b2 = n.b2; G.b3 = n.b3;
    G.D = n.D; G.H1 = n.H1; G.H2 = n.H2; G.A = n.OUT; G.out_act = a->out_act;
    G.log_var = a->log_var; G.noise_scale = a->noise_scale; G.eps = a->eps;
    G.zsum = a->zsum; G.zsumsq = a->zsumsq; G.zcount = a->zcount; G.zeps = a->zeps;
    G.state = a->state; G.init_state = a->init_state;
    G.n = a->n; G.t0 = a->t; G.episode_len = a->episode_len; G.steps = a->steps; G.R = a->rows_per_actor; G.slot0 = a->slot;
    G.obs_roll = a->obs_roll; G.act_roll = a->act_roll; G.rew_roll = a->rew_roll; G.done_roll = a->done_roll;
    G.pd_roll = a->pd_roll; G.obs_last = a->obs_last;
    // 4 or 8 actors per workgroup while that grid fits the chip once, else 16 (SMX_ROLLOUT_RG = 1 | 2 | 4 overrides: measurements)
    int rg = 4;
    for (int c = 1; c <= 2; c *= 2)
        if ((a->n + 4 * c - 1) / (4 * c) <= device_cus()) { rg = c; break; }
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("SMX_ROLLOUT_RG");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 1 || forced == 2 || forced == 4) rg = forced;
    int lds = carve(G, 4 * rg);
    // one workgroup per CU: each streams the packed weights through the CU's four SIMDs by itself
    if (lds < ROLL_EXCLUSIVE_LDS) lds = ROLL_EXCLUSIVE_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)rollout_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ROLL_MAX_LDS);
        (void)hipFuncSetAttribute((const void*)rollout_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ROLL_MAX_LDS);
        (void)hipFuncSetAttribute((const void*)rollout16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ROLL_MAX_LDS);
        attr_set = true;
    }
    const int blocks = (a->n + 4 * rg - 1) / (4 * rg);
    if (rg == 1) hipLaunchKernelGGL(rollout_kernel<1>, dim3(blocks), dim3(RNTH), lds, smx_s(stream), G);
    else if (rg == 2) hipLaunchKernelGGL(rollout_kernel<2>, dim3(blocks), dim3(RNTH), lds, smx_s(stream), G);
    else hipLaunchKernelGGL(rollout16_kernel, dim3(blocks), dim3(RNTH), lds, smx_s(stream), G);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
