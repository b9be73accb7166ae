// LSTM stem of the PPO model (surreal/model/ppo_net.py:143-152, 277-279, 338-349:
// nn.LSTM(in, hidden, 1, batch_first=True) in front of the actor / critic MLPs).
//
// Batch rows are independent, time steps are not, so the recurrence is ONE launch: a workgroup
// owns a few batch rows and walks the whole sequence with h/c resident on chip.
//   * the input half of the gates, x . W_ih^T + b_ih, has no time dependence: one GEMM over all
//     B*T rows (smx_linear_f32) before the recurrent kernel;
//   * per step the recurrent half h_{t-1} . W_hh^T with h_{t-1} read from LDS:
//     H <= 128 (the reference default is 100), B < 512: ONE row per workgroup on the vector ALU, a unit's four gates in a
//     quad of lanes with their W_hh rows in REGISTERS for the whole sequence (lstm_fwdk / lstm_bwdk_kernel; the kernels
//     they replaced remain behind SMX_LSTM_V1 / _QUAD / _MFMA4 for A/B runs); B >= 512 (H <= 112): FOUR rows per
//     workgroup on v_mfma_f32_4x4x1 (lstm_fwdm / lstm_bwdm_kernel, round 5); larger H: 16 rows on FP32 MFMA 16x16x4, W_hh
//     fragments re-read from L2 every step;
//   * the cell update is elementwise on a fixed (row, unit) -> thread map; h_t goes back to LDS.
// Backward is the mirror image (t = T-1 .. 0, dh_rec = dgates_t . W_hh on MFMA), followed by the
// weight-gradient GEMMs over all B*T rows (split-K).
#include "smx_common.h"
#include <stdlib.h>

namespace {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int RB = 16;    // batch rows per workgroup (one MFMA tile high)
constexpr int NT = 512;   // threads per workgroup
constexpr int NWV = NT / 64;

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

struct FwdArgs {
    const float* W_hh;   // [4H, H]
    const float* b_hh;   // [4H]
    const float* h0;     // [B, H] or null (zeros)
    const float* c0;
    float* gates;        // [B, T, 4H] in: x.W_ih^T + b_ih ; out: activated gates (i, f, g, o)
    float* out;          // [B, T, H]  h_t
    float* cs;           // [B, T, H]  c_t
    float* hprev;        // [B, T, H]  h_{t-1} (input of step t) or null
    float* hN;           // [B, H] final state or null
    float* cN;
    const int* stop;
    int B, T, H;
    // lstm_fwdk_kernel<.., FOLD = true>: the input half of the gates formed INSIDE the recurrence (D <= 20)
    const float* x;      // [B, T, D]
    const float* W_ih;   // [4H, D]
    const float* b_ih;   // [4H]
    int D;
};

// ---- 16-row workgroups on 16x16x4 MFMA: the large-H variant (112 < H <= 384).  W_hh does not fit
// the register file, so its fragments are re-read from L2 every step; the cell state lives in LDS.
// KG: 16-wide k groups covering H (H <= 16*KG).
template <int KG>
__global__ __launch_bounds__(NT) void lstm_fwd_kernel(FwdArgs a) {
    extern __shared__ float smem[];
    if (a.stop && *a.stop) return;
    const int H = a.H, G = 4 * H, T = a.T;
    constexpr int HS = KG * 16 + 4;            // h row stride in LDS (columns >= H stay zero)
    const int GS = ((G + 15) & ~15) + 4;       // gate row stride in LDS
    float* hs = smem;
    float* gh = smem + RB * HS;
    float* cl = gh + RB * GS;                  // [RB][H] cell state
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * RB;
    const int nct = (G + 15) >> 4;             // gate column tiles
    const int nkg = (H + 15) >> 4;

    for (int idx = tid; idx < RB * HS; idx += NT) hs[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < RB * H; idx += NT) {
        const int row = idx / H, j = idx - row * H;
        const bool inb = row0 + row < a.B;
        cl[idx] = (inb && a.c0) ? a.c0[(size_t)(row0 + row) * H + j] : 0.f;
        if (inb && a.h0) hs[row * HS + j] = a.h0[(size_t)(row0 + row) * H + j];
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        // ---- recurrent half: gh = h_{t-1} . W_hh^T + b_hh ----------------------------
        for (int ct = wv; ct < nct; ct += NWV) {
            const int col = ct * 16 + i;
            const float bias = (col < G) ? a.b_hh[col] : 0.f;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kg = 0; kg < nkg; ++kg) {
                const int k = 16 * kg + 4 * kq;
                const float4 av = *reinterpret_cast<const float4*>(&hs[i * HS + k]);
                const float4 wv4 =
                    (col < G && k < H)
                        ? *reinterpret_cast<const float4*>(a.W_hh + (size_t)col * H + k)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = MFMA16(av.x, wv4.x, acc);
                acc = MFMA16(av.y, wv4.y, acc);
                acc = MFMA16(av.z, wv4.z, acc);
                acc = MFMA16(av.w, wv4.w, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gh[(4 * kq + r) * GS + col] = acc[r] + bias;
        }
        __syncthreads();

        // ---- cell update (aten lstm_cell: gates = igates + hgates; i, f, g, o) ---------
        for (int idx = tid; idx < RB * H; idx += NT) {
            const int row = idx / H, j = idx - row * H;
            if (row0 + row < a.B) {
                const unsigned oh = ((unsigned)(row0 + row) * (unsigned)T + (unsigned)t) * (unsigned)H + j;
                const unsigned og = (oh - j) * 4u + j;
                const float* g = gh + row * GS + j;
                const float gi = sigm(a.gates[og] + g[0]);
                const float gf = sigm(a.gates[og + (unsigned)H] + g[H]);
                const float gg = tanhf(a.gates[og + (unsigned)(2 * H)] + g[2 * H]);
                const float go = sigm(a.gates[og + (unsigned)(3 * H)] + g[3 * H]);
                const float c = gf * cl[idx] + gi * gg;
                const float h = go * tanhf(c);
                a.gates[og] = gi; a.gates[og + (unsigned)H] = gf;
                a.gates[og + (unsigned)(2 * H)] = gg; a.gates[og + (unsigned)(3 * H)] = go;
                a.out[oh] = h;
                a.cs[oh] = c;
                if (a.hprev) a.hprev[oh] = hs[row * HS + j];
                cl[idx] = c;
                hs[row * HS + j] = h;
            }
        }
        __syncthreads();
    }

    for (int idx = tid; idx < RB * H; idx += NT) {
        const int row = idx / H, j = idx - row * H;
        if (row0 + row < a.B) {
            if (a.hN) a.hN[(size_t)(row0 + row) * H + j] = hs[row * HS + j];
            if (a.cN) a.cN[(size_t)(row0 + row) * H + j] = cl[idx];
        }
    }
}

struct BwdArgs {
    const float* W_hh;
    const float* c0;      // [B, H] or null
    const float* gates;   // activated gates from the forward pass
    const float* cs;
    const float* dout;    // [B, T, H] gradient w.r.t. h_t from the layers above
    float* dgates;        // [B, T, 4H] gradient w.r.t. the pre-activation gates (may alias gates)
    const int* stop;
    int B, T, H;
};

template <int KG>
__global__ __launch_bounds__(NT) void lstm_bwd_kernel(BwdArgs a) {
    extern __shared__ float smem[];
    if (a.stop && *a.stop) return;
    const int H = a.H, G = 4 * H, T = a.T;
    constexpr int HS = KG * 16 + 4;
    const int GS = ((G + 15) & ~15) + 4;
    float* dhr = smem;                 // [RB][HS]  recurrent dh
    float* dg = smem + RB * HS;        // [RB][GS]  dgates of the current step
    float* dcl = dg + RB * GS;         // [RB][H]   dc carried to step t-1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * RB;
    const int nht = (H + 15) >> 4;     // hidden-unit tiles
    const int nkg = (G + 15) >> 4;

    for (int idx = tid; idx < RB * (HS + GS + H); idx += NT) smem[idx] = 0.f;
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        for (int idx = tid; idx < RB * H; idx += NT) {
            const int row = idx / H, j = idx - row * H;
            if (row0 + row < a.B) {
                const unsigned oh = ((unsigned)(row0 + row) * (unsigned)T + (unsigned)t) * (unsigned)H + j;
                const unsigned og = (oh - j) * 4u + j;
                const float gi = a.gates[og], gf = a.gates[og + (unsigned)H];
                const float gg = a.gates[og + (unsigned)(2 * H)], go = a.gates[og + (unsigned)(3 * H)];
                const float c = a.cs[oh];
                const float cp = (t > 0) ? a.cs[oh - (unsigned)H]
                                         : (a.c0 ? a.c0[(size_t)(row0 + row) * H + j] : 0.f);
                const float dh = a.dout[oh] + dhr[row * HS + j];
                const float tc = tanhf(c);
                const float dc = dcl[idx] + (dh * go) * (1.f - tc * tc);
                const float dgi = (dc * gg) * (gi * (1.f - gi));
                const float dgf = (dc * cp) * (gf * (1.f - gf));
                const float dgg = (dc * gi) * (1.f - gg * gg);
                const float dgo = (dh * tc) * (go * (1.f - go));
                dcl[idx] = dc * gf;
                a.dgates[og] = dgi; a.dgates[og + (unsigned)H] = dgf;
                a.dgates[og + (unsigned)(2 * H)] = dgg; a.dgates[og + (unsigned)(3 * H)] = dgo;
                float* gl = dg + row * GS + j;
                gl[0] = dgi; gl[H] = dgf; gl[2 * H] = dgg; gl[3 * H] = dgo;
            }
        }
        __syncthreads();
        if (t > 0) {   // dh_rec feeds step t-1 only
            for (int ht = wv; ht < nht; ht += NWV) {
                const int n = ht * 16 + i;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int kg = 0; kg < nkg; ++kg) {
                    const int k = 16 * kg + 4 * kq;
                    const float4 av = *reinterpret_cast<const float4*>(&dg[i * GS + k]);
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n < H && k < G) {   // G % 4 == 0: the four k are in range together
                        w.x = a.W_hh[(size_t)(k + 0) * H + n];
                        w.y = a.W_hh[(size_t)(k + 1) * H + n];
                        w.z = a.W_hh[(size_t)(k + 2) * H + n];
                        w.w = a.W_hh[(size_t)(k + 3) * H + n];
                    }
                    acc = MFMA16(av.x, w.x, acc);
                    acc = MFMA16(av.y, w.y, acc);
                    acc = MFMA16(av.z, w.z, acc);
                    acc = MFMA16(av.w, w.w, acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) dhr[(4 * kq + r) * HS + n] = acc[r];
            }
        }
        __syncthreads();
    }
}

// ===========================================================================================
// 4-row formulation (H <= 112): the same recurrence on v_mfma_f32_4x4x1_16b_f32.  One instruction
// is 16 independent 4x4 outer products (K = 1): block b of lane l = 4b + j multiplies the 4 batch
// rows h[i][k] (A operand, lane & 3 = i, the same for every block) with gate column 64w + l of
// W_hh (B operand, lane = column) -- 4 rows x 64 columns per instruction at the same FLOP rate as
// the 16x16x4 form, but a workgroup now owns 4 batch rows instead of 16: four times the
// workgroups (B = 64 -> 16, B = 1024 -> 256 = the whole chip) and a quarter of the serial work
// per time step.  Wave w keeps W_hh[64w + lane][0..H) in registers for the whole sequence.
// ===========================================================================================
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
// barrier between phases that exchange data through LDS only (__syncthreads() also drains the global stores)
#define LSTM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int RB4 = 4;

// KQ: 4-wide k groups covering H (H <= 4*KQ)
template <int KQ>
__global__ __launch_bounds__(NT) void lstm_fwd4_kernel(FwdArgs a) {
    extern __shared__ float smem[];
    if (a.stop && *a.stop) return;
    const int H = a.H, G = 4 * H, T = a.T;
    constexpr int HS = KQ * 4 + 4;             // h row stride in LDS (columns >= H stay zero)
    constexpr int GS = NWV * 64 + 4;           // gate row stride (every wave writes its 64 columns)
    float* hs = smem;                          // [4][HS]
    float* gh = smem + RB4 * HS;               // [4][GS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * RB4;
    const int col = wv * 64 + lane;            // gate column of this lane

    float4 wq[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q)
        wq[q] = (col < G && 4 * q < H)
                    ? *reinterpret_cast<const float4*>(a.W_hh + (size_t)col * H + 4 * q)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bias = (col < G) ? a.b_hh[col] : 0.f;

    for (int idx = tid; idx < RB4 * HS; idx += NT) hs[idx] = 0.f;
    __syncthreads();
    // one (row, unit) element per thread
    const int erow = tid / H, ej = tid - erow * H;
    const bool ev = tid < RB4 * H;
    const bool inb = ev && row0 + erow < a.B;
    const unsigned eoff = inb ? (unsigned)(row0 + erow) * (unsigned)(T * H) + (unsigned)ej : 0u;
    float creg = (inb && a.c0) ? a.c0[(size_t)(row0 + erow) * H + ej] : 0.f;
    if (inb && a.h0) hs[erow * HS + ej] = a.h0[(size_t)(row0 + erow) * H + ej];
    __syncthreads();

    const float* hrow = hs + (lane & 3) * HS;
    for (int t = 0; t < T; ++t) {
        float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gx3 = 0.f;
        if (inb) {
            const unsigned og = (eoff - ej + (unsigned)(t * H)) * 4u + ej;
            gx0 = a.gates[og]; gx1 = a.gates[og + (unsigned)H];
            gx2 = a.gates[og + (unsigned)(2 * H)]; gx3 = a.gates[og + (unsigned)(3 * H)];
        }
        if (wv * 64 < G) {                     // wave-uniform: waves past the last gate column idle
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const float4 hv = *reinterpret_cast<const float4*>(hrow + 4 * q);
                acc = MFMA4(hv.x, wq[q].x, acc);
                acc = MFMA4(hv.y, wq[q].y, acc);
                acc = MFMA4(hv.z, wq[q].z, acc);
                acc = MFMA4(hv.w, wq[q].w, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gh[r * GS + col] = acc[r] + bias;
        }
        __syncthreads();
        if (ev) {
            const float* g = gh + erow * GS + ej;
            const float gi = sigm(gx0 + g[0]);
            const float gf = sigm(gx1 + g[H]);
            const float gg = tanhf(gx2 + g[2 * H]);
            const float go = sigm(gx3 + g[3 * H]);
            const float c = gf * creg + gi * gg;
            const float h = go * tanhf(c);
            if (inb) {
                const unsigned oh = eoff + (unsigned)(t * H);
                const unsigned og = (oh - ej) * 4u + ej;
                a.gates[og] = gi; a.gates[og + (unsigned)H] = gf;
                a.gates[og + (unsigned)(2 * H)] = gg; a.gates[og + (unsigned)(3 * H)] = go;
                a.out[oh] = h;
                a.cs[oh] = c;
                if (a.hprev) a.hprev[oh] = hs[erow * HS + ej];
                creg = c;
                hs[erow * HS + ej] = h;
            }
        }
        __syncthreads();
    }
    if (inb) {
        if (a.hN) a.hN[(size_t)(row0 + erow) * H + ej] = hs[erow * HS + ej];
        if (a.cN) a.cN[(size_t)(row0 + erow) * H + ej] = creg;
    }
}

// Backward: dh_rec[4][H] = dgates_t[4][4H] . W_hh[4H][H].  The K = 4H sum is split over the four
// gate blocks: wave w handles gate block w & 3 for hidden columns 64 (w >> 2) + lane and keeps its
// W_hh slice in registers; the four partial sums meet in LDS and are added in a fixed order by the
// next step's elementwise phase.
template <int KQ>
__global__ __launch_bounds__(NT) void lstm_bwd4_kernel(BwdArgs a) {
    extern __shared__ float smem[];
    if (a.stop && *a.stop) return;
    const int H = a.H, G = 4 * H, T = a.T;
    constexpr int DS = 4 * (KQ * 4) + 4;       // dgates row stride: four gate blocks of KQ*4 (zero padded)
    constexpr int PS = 128 + 4;                // partial row stride (two 64-column groups)
    float* dg = smem;                          // [4][DS]   block q at columns [q*KQ*4, q*KQ*4 + H)
    float* part = smem + RB4 * DS;             // [4 gate blocks][4 rows][PS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * RB4;
    const int gb = wv & 3, cg = wv >> 2;
    const int n = cg * 64 + lane;              // hidden column of this lane

    float4 wq[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < H && 4 * q < H) {
            const float* p = a.W_hh + ((size_t)gb * H + 4 * q) * H + n;
            w.x = p[0]; w.y = p[H]; w.z = p[2 * (size_t)H]; w.w = p[3 * (size_t)H];
        }
        wq[q] = w;
    }
    for (int idx = tid; idx < RB4 * DS + 4 * RB4 * PS; idx += NT) smem[idx] = 0.f;
    const int erow = tid / H, ej = tid - erow * H;
    const bool ev = tid < RB4 * H;
    const bool inb = ev && row0 + erow < a.B;
    const unsigned eoff = inb ? (unsigned)(row0 + erow) * (unsigned)(T * H) + (unsigned)ej : 0u;
    float dcreg = 0.f;
    __syncthreads();

    const float* drow = dg + (lane & 3) * DS + gb * (KQ * 4);
    for (int t = T - 1; t >= 0; --t) {
        if (inb) {
            const unsigned oh = eoff + (unsigned)(t * H);
            const unsigned og = (oh - ej) * 4u + ej;
            const float gi = a.gates[og], gf = a.gates[og + (unsigned)H];
            const float gg = a.gates[og + (unsigned)(2 * H)], go = a.gates[og + (unsigned)(3 * H)];
            const float c = a.cs[oh];
            const float cp = (t > 0) ? a.cs[oh - (unsigned)H]
                                     : (a.c0 ? a.c0[(size_t)(row0 + erow) * H + ej] : 0.f);
            const float* pp = part + erow * PS + ej;
            const float dhr = ((pp[0] + pp[RB4 * PS]) + pp[2 * RB4 * PS]) + pp[3 * RB4 * PS];
            const float dh = a.dout[oh] + dhr;
            const float tc = tanhf(c);
            const float dc = dcreg + (dh * go) * (1.f - tc * tc);
            const float dgi = (dc * gg) * (gi * (1.f - gi));
            const float dgf = (dc * cp) * (gf * (1.f - gf));
            const float dgg = (dc * gi) * (1.f - gg * gg);
            const float dgo = (dh * tc) * (go * (1.f - go));
            dcreg = dc * gf;
            a.dgates[og] = dgi; a.dgates[og + (unsigned)H] = dgf;
            a.dgates[og + (unsigned)(2 * H)] = dgg; a.dgates[og + (unsigned)(3 * H)] = dgo;
            float* gl = dg + erow * DS + ej;
            gl[0] = dgi; gl[KQ * 4] = dgf; gl[2 * KQ * 4] = dgg; gl[3 * KQ * 4] = dgo;
        }
        __syncthreads();
        if (t > 0 && cg * 64 < H) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const float4 dv = *reinterpret_cast<const float4*>(drow + 4 * q);
                acc = MFMA4(dv.x, wq[q].x, acc);
                acc = MFMA4(dv.y, wq[q].y, acc);
                acc = MFMA4(dv.z, wq[q].z, acc);
                acc = MFMA4(dv.w, wq[q].w, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(gb * RB4 + r) * PS + n] = acc[r];
        }
        __syncthreads();
    }
}

// ===========================================================================================
// ONE batch row per workgroup, on the vector ALU (H <= 128).  Training runs B sequences of N - horizon + 1 (124)
// steps 43 times per learn: at B = 64 the 4-row MFMA kernels keep 16 CUs busy for ~2.1 us per step, of which the
// matrix pipe needs ~0.9 (a v_mfma_f32_4x4x1 pass computes 4 rows whether it has them or not) and the five
// transcendentals per (row, unit) thread ~0.8.  Here thread `col` owns gate column col with W_hh[col][0..H) in
// registers: H fused multiply-adds against h_{t-1} broadcast from LDS, then ITS gate's one activation; the H unit
// threads then form c_t, h_t.  Per step: ~100 FMAs + 1 transcendental + 2 short barriers + 2 transcendentals on the
// unit threads -- and B workgroups instead of B / 4.  Same FLOP rate per row as the MFMA form (H = 100: 40 k MACs per
// row-step = 312 cycles of a CU's FP32 lanes vs 350 of its matrix pipes per row).  Measured: 14.7 -> 11.1 ms per
// learn at 64 x 128, 26.5 -> 22.8 at 256 x 128 (SMX_LSTM_MFMA4=1 selects the 4-row kernels for comparison).
// ===========================================================================================
template <int HQ>          // H <= 4 HQ
__global__ __launch_bounds__(NT) void lstm_fwd1_kernel(FwdArgs a) {
    if (a.stop && *a.stop) return;
    __shared__ float4 hs4[HQ];                 // h_{t-1}, zero padded
    __shared__ float gact[4 * 4 * HQ];         // activated gates of the step
    float* hs = reinterpret_cast<float*>(hs4);
    const int H = a.H, G = 4 * H, T = a.T;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const bool colv = tid < G, unit = tid < H;
    float4 w[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q)
        w[q] = (colv && 4 * q < H) ? *reinterpret_cast<const float4*>(a.W_hh + (size_t)tid * H + 4 * q)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bias = colv ? a.b_hh[tid] : 0.f;
    const bool is_g = tid >= 2 * H && tid < 3 * H;     // the cell candidate: tanh; the other gates: sigmoid
    if (tid < 4 * HQ) hs[tid] = (unit && a.h0) ? a.h0[(size_t)b * H + tid] : 0.f;
    float creg = (unit && a.c0) ? a.c0[(size_t)b * H + tid] : 0.f;
    const size_t gbase = (size_t)b * T * G, hbase = (size_t)b * T * H;
    float gx = colv ? a.gates[gbase + tid] : 0.f;      // the input half of step 0 (smx_linear_f32)
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const float4 hv = hs4[q];
            acc0 = __builtin_fmaf(hv.x, w[q].x, acc0);
            acc1 = __builtin_fmaf(hv.y, w[q].y, acc1);
            acc0 = __builtin_fmaf(hv.z, w[q].z, acc0);
            acc1 = __builtin_fmaf(hv.w, w[q].w, acc1);
        }
        const float pre = gx + ((acc0 + acc1) + bias);
        const float act = is_g ? tanhf(pre) : sigm(pre);
        if (colv) {
            gact[tid] = act;
            a.gates[gbase + (size_t)t * G + tid] = act;
        }
        gx = (colv && t + 1 < T) ? a.gates[gbase + (size_t)(t + 1) * G + tid] : 0.f;
        LSTM_LDS_BARRIER();
        if (unit) {
            const float gi = gact[tid], gf = gact[H + tid], gg = gact[2 * H + tid], go = gact[3 * H + tid];
            const float c = gf * creg + gi * gg;
            const float h = go * tanhf(c);
            const size_t oh = hbase + (size_t)t * H + tid;
            a.out[oh] = h;
            a.cs[oh] = c;
            if (a.hprev) a.hprev[oh] = hs[tid];
            creg = c;
            hs[tid] = h;
        }
        LSTM_LDS_BARRIER();
    }
    if (unit) {
        if (a.hN) a.hN[(size_t)b * H + tid] = hs[tid];
        if (a.cN) a.cN[(size_t)b * H + tid] = creg;
    }
}

// backward of the same: thread (gb = tid / H, n = tid % H) keeps W_hh[gb H + k][n], k < H, in registers and forms gate
// block gb's share of dh_rec[n]; the four shares are added in a fixed order by the unit threads of the next step
template <int HQ>
__global__ __launch_bounds__(NT) void lstm_bwd1_kernel(BwdArgs a) {
    if (a.stop && *a.stop) return;
    __shared__ float4 dg4[4][HQ];              // dgates of the step, gate block gb at dg4[gb] (zero padded)
    __shared__ float part[4][4 * HQ];
    const int H = a.H, G = 4 * H, T = a.T;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const bool colv = tid < G, unit = tid < H;
    const int gb = colv ? tid / H : 0, n = colv ? tid - gb * H : 0;
    float4 w[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (colv && 4 * q < H) {
            const float* p = a.W_hh + ((size_t)gb * H + 4 * q) * H + n;
            v = make_float4(p[0], p[H], p[2 * (size_t)H], p[3 * (size_t)H]);
        }
        w[q] = v;
    }
    for (int idx = tid; idx < 4 * 4 * HQ; idx += NT) {
        reinterpret_cast<float*>(dg4)[idx] = 0.f;
        reinterpret_cast<float*>(part)[idx] = 0.f;
    }
    const size_t gbase = (size_t)b * T * G, hbase = (size_t)b * T * H;
    float dcreg = 0.f;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c = 0.f, cp = 0.f, dout = 0.f;
    auto fetch = [&](int t, float& xi, float& xf, float& xg, float& xo, float& xcp, float& xd) {
        const size_t og = gbase + (size_t)t * G + tid, oh = hbase + (size_t)t * H + tid;
        xi = a.gates[og]; xf = a.gates[og + H]; xg = a.gates[og + 2 * (size_t)H]; xo = a.gates[og + 3 * (size_t)H];
        xcp = (t > 0) ? a.cs[oh - H] : (a.c0 ? a.c0[(size_t)b * H + tid] : 0.f);
        xd = a.dout[oh];
    };
    if (unit) {
        fetch(T - 1, gi, gf, gg, go, cp, dout);
        c = a.cs[hbase + (size_t)(T - 1) * H + tid];
    }
    __syncthreads();
    float* dg = reinterpret_cast<float*>(dg4);
    for (int t = T - 1; t >= 0; --t) {
        float ni = 0.f, nf = 0.f, ng = 0.f, no = 0.f, ncp = 0.f, nd = 0.f;
        if (unit && t > 0) fetch(t - 1, ni, nf, ng, no, ncp, nd);      // the next step's inputs: requested early
        if (unit) {
            const float dhr = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
            const float dh = dout + dhr;
            const float tc = tanhf(c);
            const float dc = dcreg + (dh * go) * (1.f - tc * tc);
            const float dgi = (dc * gg) * (gi * (1.f - gi));
            const float dgf = (dc * cp) * (gf * (1.f - gf));
            const float dgg = (dc * gi) * (1.f - gg * gg);
            const float dgo = (dh * tc) * (go * (1.f - go));
            dcreg = dc * gf;
            const size_t og = gbase + (size_t)t * G + tid;
            a.dgates[og] = dgi; a.dgates[og + H] = dgf;
            a.dgates[og + 2 * (size_t)H] = dgg; a.dgates[og + 3 * (size_t)H] = dgo;
            dg[tid] = dgi; dg[4 * HQ + tid] = dgf; dg[8 * HQ + tid] = dgg; dg[12 * HQ + tid] = dgo;
        }
        LSTM_LDS_BARRIER();
        if (t > 0 && colv) {
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const float4 dv = dg4[gb][q];
                acc0 = __builtin_fmaf(dv.x, w[q].x, acc0);
                acc1 = __builtin_fmaf(dv.y, w[q].y, acc1);
                acc0 = __builtin_fmaf(dv.z, w[q].z, acc0);
                acc1 = __builtin_fmaf(dv.w, w[q].w, acc1);
            }
            part[gb][n] = acc0 + acc1;
        }
        c = cp;
        gi = ni; gf = nf; gg = ng; go = no; cp = ncp; dout = nd;
        LSTM_LDS_BARRIER();
    }
}

// ===========================================================================================
// The same one-row-per-workgroup recurrence with the four gates of a hidden unit in ADJACENT LANES (thread 4 n + g:
// unit n, gate g).  What that buys per time step:
//   * the gates of a unit meet through DPP quad broadcasts -- no LDS round trip and no barrier between the gate
//     activations and the cell update, which all four lanes of a quad now form redundantly (on all eight waves instead
//     of two waves working while six wait);
//   * h_t (forward) / the step's dgates (backward) ping-pong between two LDS buffers, so ONE barrier per step is
//     enough (a buffer is rewritten two steps later, behind the barrier of the step in between);
//   * backward: a quad's four partial sums of dh_rec stay in registers and are added by DPP in the fixed order
//     ((p0 + p1) + p2) + p3 -- the LDS exchange and its barrier are gone;
//   * the H multiply-adds per lane are packed two to an instruction (v_pk_fma_f32), same two accumulation chains.
// Every value is formed by the same operations in the same order as in lstm_fwd1 / lstm_bwd1_kernel (bit-identical
// results); SMX_LSTM_V1=1 selects those for A/B runs.
// ===========================================================================================
typedef float v2f __attribute__((ext_vector_type(2)));

template <int G>
__device__ __forceinline__ float quad_bcast(float v) {      // lane G of every quad -> all four lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), G * 0x55, 0xf, 0xf, true));
}

template <int HQ>          // H <= 4 HQ
__global__ __launch_bounds__(NT) void lstm_fwdq_kernel(FwdArgs a) {
    if (a.stop && *a.stop) return;
    __shared__ float4 hs4[2][HQ];              // h_{t-1} / h_t, zero padded
    const int H = a.H, G = 4 * H, T = a.T;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const bool colv = tid < G;
    const int n = tid >> 2, g = tid & 3;
    const int col = colv ? g * H + n : 0;
    float4 w[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q)
        w[q] = (colv && 4 * q < H) ? *reinterpret_cast<const float4*>(a.W_hh + (size_t)col * H + 4 * q)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bias = colv ? a.b_hh[col] : 0.f;
    const bool is_g = g == 2;                  // the cell candidate: tanh; the other gates: sigmoid
    float* hs = reinterpret_cast<float*>(hs4);
    for (int i = tid; i < 2 * 4 * HQ; i += NT) hs[i] = 0.f;
    __syncthreads();
    if (colv && g == 0 && a.h0) hs[n] = a.h0[(size_t)b * H + n];
    float creg = (colv && a.c0) ? a.c0[(size_t)b * H + n] : 0.f;
    const size_t gbase = (size_t)b * T * G, hbase = (size_t)b * T * H;
    float gx = colv ? a.gates[gbase + col] : 0.f;      // the input half of step 0 (smx_linear_f32)
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int p = t & 1;
        v2f acc = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const float4 hv = hs4[p][q];
            acc = __builtin_elementwise_fma((v2f){hv.x, hv.y}, (v2f){w[q].x, w[q].y}, acc);
            acc = __builtin_elementwise_fma((v2f){hv.z, hv.w}, (v2f){w[q].z, w[q].w}, acc);
        }
        const float pre = gx + ((acc.x + acc.y) + bias);
        const float act = is_g ? tanhf(pre) : sigm(pre);
        if (colv) a.gates[gbase + (size_t)t * G + col] = act;
        gx = (colv && t + 1 < T) ? a.gates[gbase + (size_t)(t + 1) * G + col] : 0.f;
        const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
        const float c = gf * creg + gi * gg;
        const float h = go * tanhf(c);
        creg = c;
        if (colv && g == 0) {
            const size_t oh = hbase + (size_t)t * H + n;
            a.out[oh] = h;
            a.cs[oh] = c;
            if (a.hprev) a.hprev[oh] = hs[p * 4 * HQ + n];
            hs[(1 - p) * 4 * HQ + n] = h;
        }
        LSTM_LDS_BARRIER();
    }
    if (colv && g == 0) {
        if (a.hN) a.hN[(size_t)b * H + n] = hs[(T & 1) * 4 * HQ + n];
        if (a.cN) a.cN[(size_t)b * H + n] = creg;
    }
}

// ===========================================================================================
// Forward recurrence with the K SUM of a hidden unit split over its quad (thread 4 n + kq: unit n, k-quarter kq, all
// four gates).  In lstm_fwdq_kernel every lane reads ALL of h_{t-1} from LDS each step: 25 ds_read_b128 per lane whose
// data return -- 64 lanes x 16 bytes per instruction, broadcast or not -- is 8 cycles of the CU's one LDS pipe each:
// 8 wavefronts x 25 x 8 = 1600 cycles of a 2100-cycle step.  Here a lane reads only ITS quarter of h (7 reads, 450
// cycles per step and CU), multiplies it with the four gates' weights (the same 4 H / 4 = 100 weights per lane, in
// registers), and the quad's partial sums meet by a DPP reduce-scatter: lane kq ends with gate kq's sum in the fixed
// order (q0 + q1) + (q2 + q3), activates it, and the cell update proceeds as in lstm_fwdq_kernel.
// ===========================================================================================
// Gate non-linearities on the hardware exp2 / rcp units.  A step of the one-row recurrence lasts as long as ONE wavefront
// needs for its own in-order instruction stream (profiles/r03_pmc_lstm.json: vector ALUs 24 % busy, LDS 2 % -- nothing is
// saturated), and more than half of that stream was the libm expf / tanhf / IEEE division of sigmoid and tanh -- on
// BOTH sides of the tanh-or-sigmoid branch, since the four gates of a unit sit in one quad.  sigmoid(z) = rcp(1 +
// exp2(-z log2 e)) is 5 instructions; tanh(x) = 2 sigmoid(2 x) - 1 shares them, so a lane's gate is ONE branch-free
// sequence.  Absolute error <= 1.5e-7 (v_exp_f32 and v_rcp_f32 are 1 ulp; the argument scaling adds |z| 2^-24 relative
// to an exponent whose sensitivity s (1 - s) |z| peaks at 0.22) against the 1e-5 parity bound.
__device__ __forceinline__ float fast_sigm(float z) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z));
}
__device__ __forceinline__ float fast_tanh(float x) { return 2.f * fast_sigm(2.f * x) - 1.f; }

__device__ __forceinline__ float quad_xor1(float v) {       // lane ^ 1 inside the quad
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_xor2(float v) {       // lane ^ 2 inside the quad
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
}

// RW batch rows per workgroup (rows RW b .. RW b + RW - 1) share the weights in registers: a step's latency chain
// (LDS read -> FMAs -> quad reduce -> activation -> LDS write -> barrier, ~1700 cycles of which the FMAs are a quarter)
// is walked once for RW rows.  One row per workgroup keeps 1024 rows in FOUR rounds of 256 workgroups at 0.18 of the
// vector rate; RW = 4 is one round.  Every row's arithmetic is what RW = 1 does: results are bit-identical.
// FOLD (D <= 20, the low-dimensional observation in front of the first layer): the input half of the gates,
// x_t . W_ih^T + b_ih, is formed inside the step -- 5 more weights per gate and lane in registers (the quad's four lanes
// split the D inputs 5 / 5 / 5 / 5, zero padded), x_t staged in LDS one step ahead by 20 lanes per row from a four-step
// register ring -- instead of being written by a GEMM launch over all B T rows and read back here: at 126 976 rows that
// launch writes 203 MB for 1.7 GFLOP (210 us, profiles/r05_lstm_1024x128_kernel_stats_b.csv) and the recurrence reads
// them again.  Same products, another summation order than igates + hgates (aten lstm_cell): within fp32 rounding.
template <int Q4, int RW, bool FOLD>   // float4 words per k-quarter: H <= 16 Q4
__global__ __launch_bounds__(NT) void lstm_fwdk_kernel(FwdArgs a) {
    if (a.stop && *a.stop) return;
    __shared__ float4 hs4[RW][2][4][Q4];       // h_{t-1} / h_t by k-quarter, zero padded
    const int H = a.H, G = 4 * H, T = a.T, QS = H >> 2;     // (H % 4 == 0)
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * RW;
    const bool colv = tid < G;
    const int n = tid >> 2, kq = tid & 3;
    float4 w[4][Q4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < Q4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (colv) {
                const float* p = a.W_hh + (size_t)(g * H + n) * H + kq * QS + 4 * q;
                if (4 * q + 0 < QS) v.x = p[0];
                if (4 * q + 1 < QS) v.y = p[1];
                if (4 * q + 2 < QS) v.z = p[2];
                if (4 * q + 3 < QS) v.w = p[3];
            }
            w[g][q] = v;
        }
    const int col = colv ? kq * H + n : 0;     // the gate column this lane activates: gate kq of unit n
    const float bias = colv ? (FOLD ? a.b_hh[col] + a.b_ih[col] : a.b_hh[col]) : 0.f;
    // FOLD: W_ih[gate g of unit n][inputs 5 kq .. 5 kq + 4]
    __shared__ float4 xs4[FOLD ? RW : 1][2][4][2];       // x_{t} / x_{t+1} by input quarter: 5 words + 3 zero pads
    float wx[4][5];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            wx[g][j] = (FOLD && colv && 5 * kq + j < a.D) ? a.W_ih[(size_t)(g * H + n) * a.D + 5 * kq + j] : 0.f;
    float* xsf = reinterpret_cast<float*>(xs4);
    // staging role: lane tid < 20 RW carries input xj of row xr through the ring into LDS
    const int xr = tid / 20, xj = tid - 20 * xr;
    const bool xlane = FOLD && tid < 20 * RW && xj < a.D && b0 + xr < a.B;
    const bool xwave = FOLD && tid < 128;      // (uniform per wavefront) the waves that hold staging lanes: 20 RW <= 80
    const int xpos = (xj / 5) * 8 + (xj % 5);
    // (a lane without a staging role reads the workgroup's first word: its loads are unconditional, see SMX_FWDK_STEP)
    const float* const xrow = FOLD ? a.x + (size_t)(b0 + (xlane ? xr : 0)) * T * a.D + (xlane ? xj : 0) : nullptr;
    const int Tm = T - 1;
    float xg0 = 0.f, xg1 = 0.f, xg2 = 0.f, xg3 = 0.f;      // x_{t+1} .. x_{t+4} of this lane's (row, input)
    if (FOLD) {
        for (int i = tid; i < RW * 2 * 32; i += NT) xsf[i] = 0.f;
    }
    const bool is_g = kq == 2;                 // the cell candidate: tanh; the other gates: sigmoid
    float* hs = reinterpret_cast<float*>(hs4);
    for (int i = tid; i < RW * 2 * 4 * 4 * Q4; i += NT) hs[i] = 0.f;
    __syncthreads();
    const int pos = colv ? (n / QS) * 4 * Q4 + (n % QS) : 0;       // unit n inside the quartered layout
    // addressing: a UNIFORM row / step base (scalar registers) + a 32-bit lane offset, so that a step's loads and stores
    // carry no 64-bit vector address arithmetic (it was a fifth of the step's VALU instructions)
    bool rv[RW];                               // (uniform) the row exists
    float creg[RW];
    float* grow[RW];
    float* orow[RW];
    float* crow[RW];
    float* prow[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        rv[r] = b0 + r < a.B;
        const size_t br = rv[r] ? (size_t)(b0 + r) : (size_t)b0;         // (a missing row reads row b0: never stored)
        if (colv && kq == 0 && a.h0) hs[r * 32 * Q4 + pos] = a.h0[br * H + n];
        creg[r] = (colv && a.c0) ? a.c0[br * H + n] : 0.f;
        grow[r] = a.gates + br * T * G;                                  // this row's gates [T][4H]
        orow[r] = a.out + br * T * H;
        crow[r] = a.cs + br * T * H;
        prow[r] = a.hprev ? a.hprev + br * T * H : nullptr;
    }
    const unsigned ucol = (unsigned)col, un = (unsigned)n;
    // the input half of the gates (smx_linear_f32 wrote it) is requested FOUR steps ahead: the [B, T, 4H] buffer does not
    // stay in L2 between the GEMM and this kernel, and a request made one step ahead -- as in lstm_fwdq_kernel -- makes
    // every 0.9 us step wait for a memory round trip of about that length.  Four named registers (per row) rotated by
    // unrolling.
    float gx0[RW], gx1[RW], gx2[RW], gx3[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        gx0[r] = gx1[r] = gx2[r] = gx3[r] = 0.f;
        if (!FOLD) {                           // (every lane, from valid addresses: ucol = 0 past the gate columns)
            gx0[r] = grow[r][ucol];
            gx1[r] = (grow[r] + (size_t)(1 < T - 1 ? 1 : T - 1) * G)[ucol];
            gx2[r] = (grow[r] + (size_t)(2 < T - 1 ? 2 : T - 1) * G)[ucol];
            gx3[r] = (grow[r] + (size_t)(3 < T - 1 ? 3 : T - 1) * G)[ucol];
        }
    }
    if (FOLD) {
        __syncthreads();                       // (the zero fill of xs4 above)
        if (xwave) {
            const float x0 = xrow[0];
            xg0 = xrow[(size_t)(1 < Tm ? 1 : Tm) * a.D];
            xg1 = xrow[(size_t)(2 < Tm ? 2 : Tm) * a.D];
            xg2 = xrow[(size_t)(3 < Tm ? 3 : Tm) * a.D];
            xg3 = xrow[(size_t)(4 < Tm ? 4 : Tm) * a.D];
            if (xlane) xsf[(xr * 2 + 0) * 32 + xpos] = x0;
        }
    }
    const bool odd = (kq & 1) != 0, hi = (kq & 2) != 0;
    __syncthreads();
#define SMX_FWDK_STEP(GX, XG, TT)                                                                                        \
    if ((TT) < T) {                                                                                                    \
        const int t = (TT);                                                                                            \
        const int p = t & 1;                                                                                           \
        v2f a0[RW], a1[RW], a2[RW], a3[RW];                                                                            \
        _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                                               \
            a0[r] = (v2f){0.f, 0.f}; a1[r] = (v2f){0.f, 0.f}; a2[r] = (v2f){0.f, 0.f}; a3[r] = (v2f){0.f, 0.f};       \
        }                                                                                                              \
        _Pragma("unroll") for (int q = 0; q < Q4; ++q) {                                                               \
            _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                                           \
                const float4 hv = hs4[r][p][kq][q];                                                                    \
                const v2f lo = {hv.x, hv.y}, up = {hv.z, hv.w};                                                        \
                a0[r] = __builtin_elementwise_fma(lo, (v2f){w[0][q].x, w[0][q].y}, a0[r]);                             \
                a1[r] = __builtin_elementwise_fma(lo, (v2f){w[1][q].x, w[1][q].y}, a1[r]);                             \
                a2[r] = __builtin_elementwise_fma(lo, (v2f){w[2][q].x, w[2][q].y}, a2[r]);                             \
                a3[r] = __builtin_elementwise_fma(lo, (v2f){w[3][q].x, w[3][q].y}, a3[r]);                             \
                a0[r] = __builtin_elementwise_fma(up, (v2f){w[0][q].z, w[0][q].w}, a0[r]);                             \
                a1[r] = __builtin_elementwise_fma(up, (v2f){w[1][q].z, w[1][q].w}, a1[r]);                             \
                a2[r] = __builtin_elementwise_fma(up, (v2f){w[2][q].z, w[2][q].w}, a2[r]);                             \
                a3[r] = __builtin_elementwise_fma(up, (v2f){w[3][q].z, w[3][q].w}, a3[r]);                             \
            }                                                                                                          \
        }                                                                                                              \
        if (FOLD) {                                                                                                    \
            _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                                           \
                const float4 xa = xs4[FOLD ? r : 0][p][kq][0], xb = xs4[FOLD ? r : 0][p][kq][1];                      \
                const float xv[5] = {xa.x, xa.y, xa.z, xa.w, xb.x};                                                    \
                _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                        \
                    a0[r].x = __builtin_fmaf(xv[j], wx[0][j], a0[r].x);                                                \
                    a1[r].x = __builtin_fmaf(xv[j], wx[1][j], a1[r].x);                                                \
                    a2[r].x = __builtin_fmaf(xv[j], wx[2][j], a2[r].x);                                                \
                    a3[r].x = __builtin_fmaf(xv[j], wx[3][j], a3[r].x);                                                \
                }                                                                                                      \
            }                                                                                                          \
            /* x_{t+1} into the other buffer; the ring moves on.  The load is UNCONDITIONAL inside a wave-uniform branch (under */ \
            /* the lane mask it was waited for at the end of the masked region: a memory round trip in the step) */      \
            if (xwave) {                                                                                               \
                if (xlane) xsf[(xr * 2 + 1 - p) * 32 + xpos] = XG;                                                     \
                XG = xrow[(size_t)(t + 5 < Tm ? t + 5 : Tm) * a.D];                                                    \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                                               \
            const float p0 = a0[r].x + a0[r].y, p1 = a1[r].x + a1[r].y, p2 = a2[r].x + a2[r].y, p3 = a3[r].x + a3[r].y; \
            /* reduce-scatter over the quad.  Step 1 (partner lane ^ 1): even lanes collect gates 0 and 2, odd lanes 1, 3 */ \
            const float s_lo = (odd ? p1 : p0) + quad_xor1(odd ? p0 : p1);                                             \
            const float s_up = (odd ? p3 : p2) + quad_xor1(odd ? p2 : p3);                                             \
            /* step 2 (partner lane ^ 2): lanes 0, 1 keep gates 0, 1; lanes 2, 3 keep gates 2, 3 */                    \
            const float tot = (hi ? s_up : s_lo) + quad_xor2(hi ? s_lo : s_up);                                        \
            const float pre = GX[r] + (tot + bias);                                                                    \
            const float sg = fast_sigm(is_g ? 2.f * pre : pre);                                                      \
            const float act = is_g ? 2.f * sg - 1.f : sg;        /* tanh for the cell candidate, sigmoid otherwise */   \
            float* const gstep = grow[r] + (size_t)t * G;                /* uniform */                                  \
            if (colv && rv[r]) gstep[ucol] = act;                                                                      \
            if (!FOLD) GX[r] = (grow[r] + (size_t)(t + 4 < T - 1 ? t + 4 : T - 1) * G)[ucol];    /* (unconditional) */ \
            const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act); \
            const float c = gf * creg[r] + gi * gg;                                                                    \
            const float h = go * fast_tanh(c);                                                                         \
            creg[r] = c;                                                                                               \
            if (colv && kq == 0) {                                                                                     \
                if (rv[r]) {                                                                                           \
                    (orow[r] + (size_t)t * H)[un] = h;                                                                 \
                    (crow[r] + (size_t)t * H)[un] = c;                                                                 \
                    if (prow[r]) (prow[r] + (size_t)t * H)[un] = hs[(r * 2 + p) * 16 * Q4 + pos];                      \
                }                                                                                                      \
                hs[(r * 2 + 1 - p) * 16 * Q4 + pos] = h;                                                               \
            }                                                                                                          \
        }                                                                                                              \
        LSTM_LDS_BARRIER();                                                                                            \
    }
    for (int t0 = 0; t0 < T; t0 += 4) {
        SMX_FWDK_STEP(gx0, xg0, t0)
        SMX_FWDK_STEP(gx1, xg1, t0 + 1)
        SMX_FWDK_STEP(gx2, xg2, t0 + 2)
        SMX_FWDK_STEP(gx3, xg3, t0 + 3)
    }
#undef SMX_FWDK_STEP
    if (colv && kq == 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (rv[r]) {
                if (a.hN) a.hN[(size_t)(b0 + r) * H + n] = hs[(r * 2 + (T & 1)) * 16 * Q4 + pos];
                if (a.cN) a.cN[(size_t)(b0 + r) * H + n] = creg[r];
            }
    }
}

template <int HQ>
__global__ __launch_bounds__(NT) void lstm_bwdq_kernel(BwdArgs a) {
    if (a.stop && *a.stop) return;
    __shared__ float4 dg4[2][4][HQ];           // the step's dgates, gate block gb at dg4[p][gb] (zero padded)
    const int H = a.H, G = 4 * H, T = a.T;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const bool colv = tid < G;
    const int n = colv ? tid >> 2 : 0, gb = tid & 3;
    float4 w[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (colv && 4 * q < H) {
            const float* p = a.W_hh + ((size_t)gb * H + 4 * q) * H + n;
            v = make_float4(p[0], p[H], p[2 * (size_t)H], p[3 * (size_t)H]);
        }
        w[q] = v;
    }
    float* dg = reinterpret_cast<float*>(dg4);
    for (int idx = tid; idx < 2 * 4 * 4 * HQ; idx += NT) dg[idx] = 0.f;
    const size_t gbase = (size_t)b * T * G, hbase = (size_t)b * T * H;
    float dcreg = 0.f, share = 0.f;            // share: this lane's gate block's part of dh_rec[n] (previous step)
    float gmine = 0.f, c = 0.f, cp = 0.f, dout = 0.f;
    auto fetch = [&](int t, float& xg, float& xcp, float& xd) {
        const size_t oh = hbase + (size_t)t * H + n;
        xg = a.gates[gbase + (size_t)t * G + (size_t)gb * H + n];      // this lane's own gate of unit n
        xcp = (t > 0) ? a.cs[oh - H] : (a.c0 ? a.c0[(size_t)b * H + n] : 0.f);
        xd = a.dout[oh];
    };
    if (colv) {
        fetch(T - 1, gmine, cp, dout);
        c = a.cs[hbase + (size_t)(T - 1) * H + n];
    }
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const int p = t & 1;
        float ng = 0.f, ncp = 0.f, nd = 0.f;
        if (colv && t > 0) fetch(t - 1, ng, ncp, nd);              // the next step's inputs: requested early
        const float gi = quad_bcast<0>(gmine), gf = quad_bcast<1>(gmine), gg = quad_bcast<2>(gmine),
                    go = quad_bcast<3>(gmine);
        const float dhr = ((quad_bcast<0>(share) + quad_bcast<1>(share)) + quad_bcast<2>(share)) + quad_bcast<3>(share);
        const float dh = dout + dhr;
        const float tc = fast_tanh(c);             // (the forward pass formed h with the same function)
        const float dc = dcreg + (dh * go) * (1.f - tc * tc);
        const float dgi = (dc * gg) * (gi * (1.f - gi));
        const float dgf = (dc * cp) * (gf * (1.f - gf));
        const float dgg = (dc * gi) * (1.f - gg * gg);
        const float dgo = (dh * tc) * (go * (1.f - go));
        dcreg = dc * gf;
        const float mine = gb == 0 ? dgi : (gb == 1 ? dgf : (gb == 2 ? dgg : dgo));
        if (colv) {
            a.dgates[gbase + (size_t)t * G + (size_t)gb * H + n] = mine;
            dg[((p * 4 + gb) * 4 * HQ) + n] = mine;
        }
        LSTM_LDS_BARRIER();
        share = 0.f;
        if (t > 0 && colv) {
            v2f acc = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const float4 dv = dg4[p][gb][q];
                acc = __builtin_elementwise_fma((v2f){dv.x, dv.y}, (v2f){w[q].x, w[q].y}, acc);
                acc = __builtin_elementwise_fma((v2f){dv.z, dv.w}, (v2f){w[q].z, w[q].w}, acc);
            }
            share = acc.x + acc.y;
        }
        c = cp;
        gmine = ng; cp = ncp; dout = nd;
    }
}

// ===========================================================================================
// Backward recurrence with TWO hidden units per 8 lanes in the dh_rec product.  In lstm_bwdq_kernel lane (n, gb) reads
// the 100 dgates of gate block gb -- 25 ds_read_b128 whose data return (64 lanes x 16 bytes = 8 cycles of the CU's LDS
// pipe each, x 8 wavefronts = 1600 cycles) bounds the 2470-cycle step.  Here the eight lanes of two adjacent quads
// (units 2 p and 2 p + 1) split the K = 4 H sum eight ways -- lane e takes half (e & 1) of gate block e >> 1 -- and
// each forms the partial sums of BOTH units from the words it reads: 13 reads per lane, the same 100 weights in
// registers.  The partials meet by DPP: the two quads swap the other unit's partial (row_shl / row_shr 4), then each
// quad adds its four in the fixed order (l0 + l1) + (l2 + l3).  The element-wise half of the step is that of
// lstm_bwdq_kernel (thread 4 n + gb).
// ===========================================================================================
__device__ __forceinline__ float row_from_plus4(float v) {   // lane i <- lane i + 4 (inside a 16-lane row)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_from_minus4(float v) {  // lane i <- lane i - 4
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
}

// RW batch rows per workgroup share the weights in registers (see lstm_fwdk_kernel): bit-identical to RW = 1 per row.
template <int HH, int RW>   // float4 words per half gate block: H <= 8 HH
__global__ __launch_bounds__(NT) void lstm_bwdk_kernel(BwdArgs a) {
    if (a.stop && *a.stop) return;
    __shared__ float4 dg4[RW][2][4][2 * HH];   // the step's dgates, gate block gb at dg4[r][p][gb] (zero padded)
    const int H = a.H, G = 4 * H, T = a.T;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * RW;
    const bool colv = tid < G;
    const int n = colv ? tid >> 2 : 0, gb = tid & 3;                   // element-wise role: gate gb of unit n
    const int e = tid & 7, uA = 2 * (tid >> 3), kgb = e >> 1;          // product role: k-eighth e of units uA, uA + 1
    const int nq = H >> 2, half0 = (nq + 1) >> 1;                      // float4 words per gate block / in its first half
    const int q0 = (e & 1) ? half0 : 0, qn = (e & 1) ? nq - half0 : half0;
    // (all loads first, from addresses valid in every lane, masked afterwards: a select or a lane mask right at a load makes
    // hipcc wait for it there -- 13 dependent round trips in front of the first step)
    float4 wA[HH], wB[HH];
#pragma unroll
    for (int q = 0; q < HH; ++q) {
        // (qn == 0 -- H == 4, odd e -- has no word of its own: row 0 of the gate block, masked below, keeps the load in bounds)
        const float* p = a.W_hh + ((size_t)kgb * H + 4 * (q < qn ? q0 + q : 0)) * H + (colv ? uA : 0);
        wA[q] = make_float4(p[0], p[H], p[2 * (size_t)H], p[3 * (size_t)H]);
        wB[q] = make_float4(p[1], p[H + 1], p[2 * (size_t)H + 1], p[3 * (size_t)H + 1]);
    }
#pragma unroll
    for (int q = 0; q < HH; ++q)
        if (!(colv && q < qn)) { wA[q] = make_float4(0.f, 0.f, 0.f, 0.f); wB[q] = wA[q]; }
    float* dg = reinterpret_cast<float*>(dg4);
    for (int idx = tid; idx < RW * 2 * 4 * 8 * HH; idx += NT) dg[idx] = 0.f;
    // addressing: uniform row / step bases + 32-bit lane offsets (no 64-bit vector address arithmetic inside the step)
    bool rv[RW];
    const float* grow[RW];
    float* dgrow[RW];
    const float* crow[RW];
    const float* drow[RW];
    const float* c0row[RW];
    float dcreg[RW], dhr[RW], gmine[RW], c[RW], cp[RW], dout[RW];      // dhr: dh_rec[n] from the previous step's product
    const unsigned ug = (unsigned)(gb * H + n), un = (unsigned)n;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        rv[r] = b0 + r < a.B;
        const size_t br = rv[r] ? (size_t)(b0 + r) : (size_t)b0;         // (a missing row reads row b0: never stored)
        grow[r] = a.gates + br * T * G;
        dgrow[r] = a.dgates + br * T * G;
        crow[r] = a.cs + br * T * H;
        drow[r] = a.dout + br * T * H;
        c0row[r] = a.c0 ? a.c0 + br * H : crow[r];       // (no initial state: any valid word, the use site takes 0)
        dcreg[r] = 0.f; dhr[r] = 0.f; gmine[r] = 0.f; c[r] = 0.f; cp[r] = 0.f; dout[r] = 0.f;
    }
    // The step's inputs are requested TWO steps ahead and UNCONDITIONALLY -- every lane from a valid address (a lane past the
    // gate columns reads column 0, a step before the sequence reads step 0), a value nobody needs is simply not used.  Under
    // `if (colv && t > 0)` the loads were waited for at the END OF THE MASKED REGION: a memory round trip inside every step
    // (measured on the 4-row kernels below: 242 -> 185 us per launch).
#define SMX_BWDK_FETCH(r, t, xg, xcp, xd)                                                                            \
    do {                                                                                                             \
        const int tt_ = (t) > 0 ? (t) : 0;                                                                           \
        xg = (grow[r] + (size_t)tt_ * G)[ug];                            /* this lane's own gate of unit n */          \
        xcp = (tt_ > 0 ? crow[r] + (size_t)(tt_ - 1) * H : c0row[r])[un];                                            \
        xd = (drow[r] + (size_t)tt_ * H)[un];                                                                        \
    } while (0)
    float mg[RW], mcp[RW], md[RW];             // step t - 1's inputs
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        SMX_BWDK_FETCH(r, T - 1, gmine[r], cp[r], dout[r]);
        c[r] = (crow[r] + (size_t)(T - 1) * H)[un];
        SMX_BWDK_FETCH(r, T - 2, mg[r], mcp[r], md[r]);
    }
    const bool upper = (tid & 4) != 0;         // the quad of unit uA + 1
    const bool have_c0 = a.c0 != nullptr;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RW; ++r)               // (loaded before the loop = in its register here: see lstm_fwdm_kernel)
        asm volatile("" : "+v"(gmine[r]), "+v"(cp[r]), "+v"(dout[r]), "+v"(c[r]), "+v"(mg[r]), "+v"(mcp[r]), "+v"(md[r]));
    for (int t = T - 1; t >= 0; --t) {
        const int p = t & 1;
        float ng[RW], ncp[RW], nd[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) SMX_BWDK_FETCH(r, t - 2, ng[r], ncp[r], nd[r]);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float gi = quad_bcast<0>(gmine[r]), gf = quad_bcast<1>(gmine[r]), gg = quad_bcast<2>(gmine[r]),
                        go = quad_bcast<3>(gmine[r]);
            const float dh = dout[r] + dhr[r];
            const float tc = fast_tanh(c[r]);          // (the forward pass formed h with the same function)
            const float dc = dcreg[r] + (dh * go) * (1.f - tc * tc);
            const float dgi = (dc * gg) * (gi * (1.f - gi));
            const float dgf = (dc * ((t > 0 || have_c0) ? cp[r] : 0.f)) * (gf * (1.f - gf));
            const float dgg = (dc * gi) * (1.f - gg * gg);
            const float dgo = (dh * tc) * (go * (1.f - go));
            dcreg[r] = dc * gf;
            const float mine = gb == 0 ? dgi : (gb == 1 ? dgf : (gb == 2 ? dgg : dgo));
            if (colv) {
                if (rv[r]) (dgrow[r] + (size_t)t * G)[ug] = mine;
                dg[(((r * 2 + p) * 4 + gb) * 8 * HH) + n] = mine;
            }
        }
        LSTM_LDS_BARRIER();
        if (t > 0) {                               // (every lane: lanes past 4 H hold zero weights)
            v2f accA[RW], accB[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) { accA[r] = (v2f){0.f, 0.f}; accB[r] = (v2f){0.f, 0.f}; }
#pragma unroll
            for (int q = 0; q < HH; ++q) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const float4 dv = dg4[r][p][kgb][q0 + q];
                    const v2f lo = {dv.x, dv.y}, up = {dv.z, dv.w};
                    accA[r] = __builtin_elementwise_fma(lo, (v2f){wA[q].x, wA[q].y}, accA[r]);
                    accB[r] = __builtin_elementwise_fma(lo, (v2f){wB[q].x, wB[q].y}, accB[r]);
                    accA[r] = __builtin_elementwise_fma(up, (v2f){wA[q].z, wA[q].w}, accA[r]);
                    accB[r] = __builtin_elementwise_fma(up, (v2f){wB[q].z, wB[q].w}, accB[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const float pA = accA[r].x + accA[r].y, pB = accB[r].x + accB[r].y;
                // the two quads swap the partial of the unit the OTHER one owns
                const float give = upper ? pA : pB;
                // (both DPP moves run with every lane enabled: under a divergent branch a disabled source lane reads as 0)
                const float from_lo = row_from_minus4(give), from_up = row_from_plus4(give);
                const float got = upper ? from_lo : from_up;
                const float s1 = (upper ? pB : pA) + got;
                const float s2 = s1 + quad_xor1(s1);
                dhr[r] = s2 + quad_xor2(s2);
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            c[r] = cp[r];
            gmine[r] = mg[r]; cp[r] = mcp[r]; dout[r] = md[r];
            mg[r] = ng[r]; mcp[r] = ncp[r]; md[r] = nd[r];
        }
    }
#undef SMX_BWDK_FETCH
}

// ===========================================================================================
// Round 5: FOUR batch rows per workgroup on v_mfma_f32_4x4x1_16B (H <= 112), for B >= 1024 -- where every CU holds a
// workgroup, the recurrence is throughput-, not latency-bound, and the vector kernels above spend 0.45 of a step issuing
// multiply-adds (lstm_fwdk / lstm_bwdk at four rows: 2.7 - 2.9 us per step and CU).  One v_mfma_f32_4x4x1 is 16
// independent 4 x 4 outer products: lane 4 b + i supplies row i of block b's A, lane 4 b + j column j of its B, and holds
// column j of its 4 x 4 result.  Round 2's 4-row kernels (lstm_fwd4 / lstm_bwd4_kernel above, SMX_LSTM_MFMA4=1) used it with
// gate columns in lane order; their results met in LDS, with two barriers and libm's expf / tanhf per step.  Here the
// blocks are laid out so that NOTHING crosses wavefronts inside a step:
//   forward   block b of wave w = hidden unit 16 w + b, its columns j = the unit's four gates (B = W_hh[j H + unit][k] in
//             registers, A = h_{t-1}[i][k] from LDS, the same for every block).  A lane then holds gate j of its unit for
//             the 4 rows, activates them (hardware exp2 / rcp), the quad exchanges gates by DPP, and lane j forms c_t, h_t
//             of ROW j -- 16 units x 4 rows = the wave's 64 lanes.  h_t goes to the other LDS buffer: ONE barrier per step.
//             FOLD (D <= 20): the input half of the gates is D more k-steps against x_t from LDS (W_ih in registers).
//   backward  dh_rec = dgates . W_hh: block b = (gate block gb = b >> 2, column quad c = b & 3): A = dgates_t[i][gb H + k]
//             from LDS, B = W_hh[gb H + k][16 w + 4 c + j]; the four gate blocks' partial sums of a column sit 16 lanes
//             apart and meet by two cross-lane adds; lane (gb, c, j) then does the element-wise step of ROW gb of its
//             column.  The step's inputs are requested one step ahead.  ONE barrier per step.
// Four accumulators per lane (k mod 4) instead of one chain of 100 dependent MFMAs, added (x + y) + (z + w).
// ===========================================================================================
// acc += A-row (LDS, 4 KQ floats) x the lane's weights, k mod 4 on four accumulators.  The operand words are requested a GROUP
// ahead (five 16-byte words = 20 MFMAs = 160 cycles of matrix pipe, more than an LDS round trip with seven wavefronts
// reading): left to itself hipcc keeps ONE word in flight (s_waitcnt lgkmcnt(1) behind every read) and the pipe waits for
// LDS in every iteration -- MFMA busy 39 % in the first version of these kernels.
template <int KQ>
__device__ __forceinline__ void mrows_product(const float* __restrict__ row, const float4 (&w)[KQ], f32x4& ax, f32x4& ay,
                                              f32x4& az, f32x4& aw) {
    constexpr int GS = 5, NG = (KQ + GS - 1) / GS;
    float4 c0, c1, c2, c3, c4, n0, n1, n2, n3, n4;
    c0 = c1 = c2 = c3 = c4 = n0 = n1 = n2 = n3 = n4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define SMX_MR_LD(dst, q) do { if ((q) < KQ) dst = *reinterpret_cast<const float4*>(row + 4 * (q)); } while (0)
#define SMX_MR_MM(src, q)                                                                    \
    do {                                                                                     \
        if ((q) < KQ) {                                                                      \
            ax = MFMA4(src.x, w[(q) < KQ ? (q) : 0].x, ax);                                  \
            ay = MFMA4(src.y, w[(q) < KQ ? (q) : 0].y, ay);                                  \
            az = MFMA4(src.z, w[(q) < KQ ? (q) : 0].z, az);                                  \
            aw = MFMA4(src.w, w[(q) < KQ ? (q) : 0].w, aw);                                  \
        }                                                                                    \
    } while (0)
    SMX_MR_LD(c0, 0); SMX_MR_LD(c1, 1); SMX_MR_LD(c2, 2); SMX_MR_LD(c3, 3); SMX_MR_LD(c4, 4);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int qn = (gi + 1) * GS, qc = gi * GS;
        __builtin_amdgcn_sched_barrier(0);
        SMX_MR_LD(n0, qn); SMX_MR_LD(n1, qn + 1); SMX_MR_LD(n2, qn + 2); SMX_MR_LD(n3, qn + 3); SMX_MR_LD(n4, qn + 4);
        __builtin_amdgcn_sched_barrier(0);
        SMX_MR_MM(c0, qc); SMX_MR_MM(c1, qc + 1); SMX_MR_MM(c2, qc + 2); SMX_MR_MM(c3, qc + 3); SMX_MR_MM(c4, qc + 4);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3; c4 = n4;
    }
#undef SMX_MR_LD
#undef SMX_MR_MM
}

template <int KQ, bool FOLD>      // H <= 4 KQ <= 112
__global__ __launch_bounds__(NT) void lstm_fwdm_kernel(FwdArgs a) {
    if (a.stop && *a.stop) return;
    constexpr int HS = KQ * 4 + 4;             // h row stride in LDS (columns >= H stay zero)
    constexpr int XS = 24 + 4;                 // x row stride (D <= 20, zero padded)
    __shared__ float hs[2][4][HS];
    __shared__ float xs[2][4][XS];
    const int H = a.H, G = 4 * H, T = a.T, D = a.D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * 4;
    const int g = lane & 3, unit = 16 * wv + (lane >> 2);
    const bool uv = unit < H;
    const bool wave_on = 16 * wv < H;          // (uniform) this wave has units
    const int col = uv ? g * H + unit : 0;     // gate column of this lane
    // (all loads first, from addresses valid in every lane; what a lane must not see is zeroed afterwards -- a select right
    // behind each load makes hipcc wait for them one by one)
    float4 wq[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q)
        wq[q] = *reinterpret_cast<const float4*>(a.W_hh + (size_t)col * H + (4 * q < H ? 4 * q : 0));
    float wis[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) wis[k] = FOLD ? a.W_ih[(size_t)col * D + (k < D ? k : 0)] : 0.f;
    const float bh = a.b_hh[col], bi = FOLD ? a.b_ih[col] : 0.f;
#pragma unroll
    for (int q = 0; q < KQ; ++q)
        if (!(uv && 4 * q < H)) wq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wi[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (FOLD && uv && 4 * q + j < D) ? wis[4 * q + j] : 0.f;
        wi[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
    const float bias = uv ? bh + bi : 0.f;
    const bool is_g = g == 2;                  // the cell candidate: tanh; the other gates: sigmoid
    // element-wise role: row g of this lane's unit
    const bool inb = uv && row0 + g < a.B;
    const size_t erow = (size_t)(inb ? row0 + g : row0);
    float creg = (inb && a.c0) ? a.c0[erow * H + unit] : 0.f;
    float* const orow = a.out + erow * T * H;
    float* const crow = a.cs + erow * T * H;
    float* const prow = a.hprev ? a.hprev + erow * T * H : nullptr;
    for (int i = tid; i < 2 * 4 * HS; i += NT) (&hs[0][0][0])[i] = 0.f;
    for (int i = tid; i < 2 * 4 * XS; i += NT) (&xs[0][0][0])[i] = 0.f;
    __syncthreads();
    if (inb && a.h0) hs[0][g][unit] = a.h0[erow * H + unit];
    // the gates buffer of the four rows (rows past the batch read row0's: never stored)
    bool rv[4];
    float* grow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rv[r] = row0 + r < a.B;
        grow[r] = a.gates + (size_t)(rv[r] ? row0 + r : row0) * T * G;
    }
    const unsigned ucol = (unsigned)col, un = (unsigned)unit;
    // not FOLD: the input half (smx_linear_f32 wrote it), requested four steps ahead (see lstm_fwdk_kernel)
    float gx0[4], gx1[4], gx2[4], gx3[4];
    // (every load of the ring is UNCONDITIONAL, from an address that is valid in every lane -- a lane without a unit reads
    // column 0, a step past the sequence reads the last one -- and a value nobody needs is simply not used: a load under
    // a lane mask is waited for at the end of the masked region, a select behind a load right there, and either puts a
    // memory round trip into every step)
    const int Tm = T - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        gx0[r] = gx1[r] = gx2[r] = gx3[r] = 0.f;
        if (!FOLD) {
            gx0[r] = grow[r][ucol];
            gx1[r] = (grow[r] + (size_t)(1 < Tm ? 1 : Tm) * G)[ucol];
            gx2[r] = (grow[r] + (size_t)(2 < Tm ? 2 : Tm) * G)[ucol];
            gx3[r] = (grow[r] + (size_t)(3 < Tm ? 3 : Tm) * G)[ucol];
        }
    }
    // FOLD staging role: lane tid < 80 carries input xj of row xr through a four-step ring into LDS
    const int xr = tid / 20, xj = tid - 20 * xr;
    const bool xlane = FOLD && tid < 80 && xj < D && row0 + xr < a.B;
    const bool xwave = FOLD && wv < 2;         // (uniform) the waves that hold staging lanes
    const float* const xrow = FOLD ? a.x + (size_t)(row0 + (xlane ? xr : 0)) * T * D + (xlane ? xj : 0) : nullptr;
    float xg0 = 0.f, xg1 = 0.f, xg2 = 0.f, xg3 = 0.f;
    if (xwave) {
        const float x0 = xrow[0];
        xg0 = xrow[(size_t)(1 < Tm ? 1 : Tm) * D];
        xg1 = xrow[(size_t)(2 < Tm ? 2 : Tm) * D];
        xg2 = xrow[(size_t)(3 < Tm ? 3 : Tm) * D];
        xg3 = xrow[(size_t)(4 < Tm ? 4 : Tm) * D];
        if (xlane) xs[0][xr][xj] = x0;
    }
    __syncthreads();
    // everything loaded so far is in its register HERE: redefined by an empty asm statement, or hipcc carries "may still be in
    // flight" for these registers around the loop's back edge and drains the memory queue (s_waitcnt vmcnt(0): this step's
    // stores and the ring's newest load) at their first use in the loop body
#pragma unroll
    for (int q = 0; q < KQ; ++q) asm volatile("" : "+v"(wq[q].x), "+v"(wq[q].y), "+v"(wq[q].z), "+v"(wq[q].w));
#pragma unroll
    for (int q = 0; q < 5; ++q) asm volatile("" : "+v"(wi[q].x), "+v"(wi[q].y), "+v"(wi[q].z), "+v"(wi[q].w));
    asm volatile("" : "+v"(creg), "+v"(xg0), "+v"(xg1), "+v"(xg2), "+v"(xg3));
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(gx0[r]), "+v"(gx1[r]), "+v"(gx2[r]), "+v"(gx3[r]));
#define SMX_FWDM_STEP(GX, XG, TT)                                                                                        \
    if ((TT) < T) {                                                                                                    \
        const int t = (TT);                                                                                            \
        const int p = t & 1;                                                                                           \
        float act[4] = {0.f, 0.f, 0.f, 0.f};                                                                           \
        /* x_{t+1} into the other buffer and the ring moves on -- at the TOP of the step: the wait for the ring's oldest load */ \
        /* that hipcc puts here (vmcnt(0): it cannot count the divergent stores in between) then meets an empty queue */ \
        if (xwave) {                                                                                                   \
            if (xlane) xs[1 - p][xr][xj] = XG;                                                                         \
            XG = xrow[(size_t)(t + 5 < Tm ? t + 5 : Tm) * D];                                                          \
        }                                                                                                              \
        if (wave_on) {                                                                                                 \
            f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ay = ax, az = ax, aw = ax;                                                \
            if (FOLD) mrows_product<5>(&xs[p][g][0], wi, ax, ay, az, aw);                                              \
            mrows_product<KQ>(&hs[p][g][0], wq, ax, ay, az, aw);                                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                            \
                const float pre = GX[r] + (((ax[r] + ay[r]) + (az[r] + aw[r])) + bias);                                \
                const float sg = fast_sigm(is_g ? 2.f * pre : pre);                                                  \
                act[r] = is_g ? 2.f * sg - 1.f : sg;                                                                   \
                float* const gstep = grow[r] + (size_t)t * G;                                                          \
                if (uv && rv[r]) gstep[ucol] = act[r];                                                                 \
                if (!FOLD) GX[r] = (grow[r] + (size_t)(t + 4 < Tm ? t + 4 : Tm) * G)[ucol];                            \
            }                                                                                                          \
        }                                                                                                              \
        /* the four gates of row g of this unit: lane j of the quad holds gate j of every row */                       \
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;                                                                  \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                \
            const float vi = quad_bcast<0>(act[r]), vf = quad_bcast<1>(act[r]), vg = quad_bcast<2>(act[r]),            \
                        vo = quad_bcast<3>(act[r]);                                                                    \
            if (g == r) { gi = vi; gf = vf; gg = vg; go = vo; }                                                        \
        }                                                                                                              \
        const float c = gf * creg + gi * gg;                                                                           \
        const float h = go * fast_tanh(c);                                                                             \
        creg = c;                                                                                                      \
        if (inb) {                                                                                                     \
            (orow + (size_t)t * H)[un] = h;                                                                            \
            (crow + (size_t)t * H)[un] = c;                                                                            \
            if (prow) (prow + (size_t)t * H)[un] = hs[p][g][unit];                                                     \
            hs[1 - p][g][unit] = h;                                                                                    \
        }                                                                                                              \
        LSTM_LDS_BARRIER();                                                                                            \
    }
    for (int t0 = 0; t0 < T; t0 += 4) {
        SMX_FWDM_STEP(gx0, xg0, t0)
        SMX_FWDM_STEP(gx1, xg1, t0 + 1)
        SMX_FWDM_STEP(gx2, xg2, t0 + 2)
        SMX_FWDM_STEP(gx3, xg3, t0 + 3)
    }
#undef SMX_FWDM_STEP
    if (inb) {
        if (a.hN) a.hN[erow * H + unit] = hs[T & 1][g][unit];
        if (a.cN) a.cN[erow * H + unit] = creg;
    }
}

template <int KQ>                 // H <= 4 KQ <= 112
__global__ __launch_bounds__(NT) void lstm_bwdm_kernel(BwdArgs a) {
    if (a.stop && *a.stop) return;
    constexpr int KB = KQ * 4;                 // a gate block's stride in the dgates tile (zero padded past H)
    constexpr int DS = 4 * KB + 4;
    __shared__ float dg[2][4][DS];             // the step's dgates: [row][gate block gb at gb KB]
    const int H = a.H, G = 4 * H, T = a.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * 4;
    const int gb = lane >> 4, j = lane & 3;
    const int n = 16 * wv + (lane & 15);       // hidden column of this lane: 16 w + 4 c + j
    const bool nv = n < H;
    const bool wave_on = 16 * wv < H;
    float4 wq[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {             // (all loads first, from valid addresses; masked afterwards)
        const float* p = a.W_hh + ((size_t)gb * H + (4 * q < H ? 4 * q : 0)) * H + (nv ? n : 0);
        wq[q] = make_float4(p[0], p[H], p[2 * (size_t)H], p[3 * (size_t)H]);
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q)
        if (!(nv && 4 * q < H)) wq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < 2 * 4 * DS; i += NT) (&dg[0][0][0])[i] = 0.f;
    // element-wise role: row gb of column n
    const bool inb = nv && row0 + gb < a.B;
    const size_t erow = (size_t)(inb ? row0 + gb : row0);
    const float* const grow = a.gates + erow * T * G;
    float* const dgrow = a.dgates + erow * T * G;
    const float* const crow = a.cs + erow * T * H;
    const float* const drow = a.dout + erow * T * H;
    const unsigned un = (unsigned)(nv ? n : 0);
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c = 0.f, cp = 0.f, dout = 0.f, dcreg = 0.f, dhr = 0.f;
    // the step's inputs are requested TWO steps ahead (they come from HBM: the forward pass wrote them a launch ago) and
    // UNCONDITIONALLY, every lane from a valid address (see lstm_fwdm_kernel)
    // (c_{-1}: the initial cell state, or -- without one -- any valid word: the use site takes 0 then)
    const float* const c0row = a.c0 ? a.c0 + erow * H : crow;
#define SMX_BWDM_FETCH(t, xi, xf, xg, xo, xcp, xd)                                                                     \
    do {                                                                                                             \
        const int tt_ = (t) > 0 ? (t) : 0;                                                                           \
        const float* gp_ = grow + (size_t)tt_ * G;                                                                   \
        xi = gp_[un]; xf = (gp_ + H)[un]; xg = (gp_ + 2 * H)[un]; xo = (gp_ + 3 * (size_t)H)[un];                    \
        xcp = (tt_ > 0 ? crow + (size_t)(tt_ - 1) * H : c0row)[un];                                                  \
        xd = (drow + (size_t)tt_ * H)[un];                                                                           \
    } while (0)
    float mi = 0.f, mf = 0.f, mg = 0.f, mo = 0.f, mcp = 0.f, md = 0.f;       // step t - 1's inputs
    SMX_BWDM_FETCH(T - 1, gi, gf, gg, go, cp, dout);
    c = (crow + (size_t)(T - 1) * H)[un];
    SMX_BWDM_FETCH(T - 2, mi, mf, mg, mo, mcp, md);
    __syncthreads();
    // (loaded before the loop = in its register here: see lstm_fwdm_kernel)
#pragma unroll
    for (int q = 0; q < KQ; ++q) asm volatile("" : "+v"(wq[q].x), "+v"(wq[q].y), "+v"(wq[q].z), "+v"(wq[q].w));
    asm volatile("" : "+v"(gi), "+v"(gf), "+v"(gg), "+v"(go), "+v"(cp), "+v"(dout), "+v"(c));
    asm volatile("" : "+v"(mi), "+v"(mf), "+v"(mg), "+v"(mo), "+v"(mcp), "+v"(md));
    for (int t = T - 1; t >= 0; --t) {
        const int p = t & 1;
        float ni, nf, ng, no, ncp, nd;
        SMX_BWDM_FETCH(t - 2, ni, nf, ng, no, ncp, nd);
        {
            const float dh = dout + dhr;
            const float tc = fast_tanh(c);             // (the forward pass formed h with the same function)
            const float dc = dcreg + (dh * go) * (1.f - tc * tc);
            const float dgi = (dc * gg) * (gi * (1.f - gi));
            const float cpu = (t > 0 || a.c0) ? cp : 0.f;      // c_{t-1}
            const float dgf = (dc * cpu) * (gf * (1.f - gf));
            const float dgg = (dc * gi) * (1.f - gg * gg);
            const float dgo = (dh * tc) * (go * (1.f - go));
            dcreg = dc * gf;
            if (inb) {
                float* dp = dgrow + (size_t)t * G;
                dp[un] = dgi; (dp + H)[un] = dgf; (dp + 2 * H)[un] = dgg; (dp + 3 * (size_t)H)[un] = dgo;
                float* dl = &dg[p][gb][n];
                dl[0] = dgi; dl[KB] = dgf; dl[2 * KB] = dgg; dl[3 * KB] = dgo;
            }
        }
        LSTM_LDS_BARRIER();
        if (t > 0 && wave_on) {
            f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ay = ax, az = ax, aw = ax;
            mrows_product<KQ>(&dg[p][j][gb * KB], wq, ax, ay, az, aw);
            // the four gate blocks' partial sums of a column sit 16 lanes apart: (p + p^16) + (. ^32) -- the same two
            // sums in every lane (a + b = b + a), so all four copies are bit-identical
            float tot[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = (ax[r] + ay[r]) + (az[r] + aw[r]);
                const float s2 = s + __shfl_xor(s, 16, 64);
                tot[r] = s2 + __shfl_xor(s2, 32, 64);
            }
            dhr = gb == 0 ? tot[0] : (gb == 1 ? tot[1] : (gb == 2 ? tot[2] : tot[3]));
        }
        c = cp;
        gi = mi; gf = mf; gg = mg; go = mo; cp = mcp; dout = md;
        mi = ni; mf = nf; mg = ng; mo = no; mcp = ncp; md = nd;
    }
#undef SMX_BWDM_FETCH
}

constexpr int KQ4 = 28;          // 4-row kernels: H <= 112

inline size_t lds4_fwd(int kq) { return (size_t)RB4 * ((kq * 4 + 4) + (NWV * 64 + 4)) * sizeof(float); }
inline size_t lds4_bwd(int kq) { return (size_t)(RB4 * (16 * kq + 4) + 4 * RB4 * (128 + 4)) * sizeof(float); }

constexpr int KG_MAX = 24;       // 16-row kernels: H <= 384 (LDS-bound)

inline size_t lds_bytes(int kg, int H) {
    const int HS = kg * 16 + 4, GS = ((4 * H + 15) & ~15) + 4;
    return (size_t)RB * (HS + GS + H) * sizeof(float);
}

}  // namespace

extern "C" int smx_lstm_forward_f32(const smx_lstm_t* net, const float* x, int64_t B, int32_t T,
                                    const float* h0, const float* c0, float* gates, float* out,
                                    float* cs, float* hprev, float* hN, float* cN,
                                    const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(net && x && gates && out && cs, SMX_E_NULL);
    SMX_REQUIRE(net->W_ih && net->W_hh && net->b_ih && net->b_hh, SMX_E_NULL);
    const int H = net->H, D = net->D;
    SMX_REQUIRE(B > 0 && T > 0 && H > 0 && D > 0 && B * T * 4 * H < (1ll << 31), SMX_E_SHAPE);
    SMX_REQUIRE(H % 4 == 0 && H <= 16 * KG_MAX, SMX_E_UNSUPPORTED);
    static const bool mfma4 = getenv("SMX_LSTM_MFMA4") != nullptr;
    static const bool v1 = getenv("SMX_LSTM_V1") != nullptr;       // the LDS-exchange one-row kernels, for A/B runs
    static const bool quad = getenv("SMX_LSTM_QUAD") != nullptr;   // every lane reads all of h (the round-3 first form)
    static const bool nofold = getenv("SMX_LSTM_NOFOLD") != nullptr;
    // D <= 20 on the default H <= 112 kernel: the input half of the gates is formed inside the recurrence (FOLD);
    // else for every (b, t) at once by a GEMM launch in front of it
    const bool fold = !nofold && !mfma4 && !v1 && !quad && H <= 112 && D <= 20;
    if (!fold) {
        int rc = smx_linear_f32(x, D, 1, net->W_ih, D, 1, net->b_ih, gates, 4 * H, (int32_t)(B * T),
                                4 * H, D, SMX_ACT_NONE, nullptr, stop_flag, stream);
        if (rc) return rc;
    }
    FwdArgs a;
    a.W_hh = net->W_hh; a.b_hh = net->b_hh; a.h0 = h0; a.c0 = c0; a.gates = gates; a.out = out;
    a.cs = cs; a.hprev = hprev; a.hN = hN; a.cN = cN; a.stop = stop_flag;
    a.B = (int)B; a.T = T; a.H = H;
    a.x = x; a.W_ih = net->W_ih; a.b_ih = net->b_ih; a.D = D;
    const int blocks = (int)((B + RB - 1) / RB);
    // H <= 128: one row per workgroup on the vector ALU (SMX_LSTM_MFMA4=1 keeps the 4-row MFMA kernels for A/B runs)
    // B >= 512: four rows per workgroup on the matrix pipes (lstm_fwdm_kernel; measured against the vector kernels at
    // 128 steps: 23.2 vs 23.7 ms per learn at 512 sequences, 28.7 vs 38.3 at 768, 34.0 vs 39.2 at 1024 -- and 17.7 vs 13.8
    // at 256, where one row per workgroup fills the chip).  SMX_LSTM_NO_MROWS=1 keeps the vector kernels for A/B runs
    static const bool no_mrows = getenv("SMX_LSTM_NO_MROWS") != nullptr;
    static const long mrows_min = getenv("SMX_LSTM_MROWS_MIN") ? atol(getenv("SMX_LSTM_MROWS_MIN")) : 512;   // (measurements)
    const bool mrows = !no_mrows && !mfma4 && !v1 && !quad && H <= 112 && B >= mrows_min;
    if (mrows) {
        const dim3 grid((unsigned)((B + 3) / 4));
        if (fold && H <= 100) hipLaunchKernelGGL((lstm_fwdm_kernel<25, true>), grid, dim3(NT), 0, smx_s(stream), a);
        else if (fold) hipLaunchKernelGGL((lstm_fwdm_kernel<28, true>), grid, dim3(NT), 0, smx_s(stream), a);
        else if (H <= 100) hipLaunchKernelGGL((lstm_fwdm_kernel<25, false>), grid, dim3(NT), 0, smx_s(stream), a);
        else hipLaunchKernelGGL((lstm_fwdm_kernel<28, false>), grid, dim3(NT), 0, smx_s(stream), a);
    } else if (fold) {
        // (four rows per workgroup with the 20 extra weights: 256 registers and spills -- two it is)
        if (B >= 512) hipLaunchKernelGGL((lstm_fwdk_kernel<7, 2, true>), dim3((unsigned)((B + 1) / 2)), dim3(NT), 0, smx_s(stream), a);
        else hipLaunchKernelGGL((lstm_fwdk_kernel<7, 1, true>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && !quad && H <= 112) {
        // (B >= 512 reaches this only with SMX_LSTM_NO_MROWS: two rows per workgroup)
        if (B >= 512) hipLaunchKernelGGL((lstm_fwdk_kernel<7, 2, false>), dim3((unsigned)((B + 1) / 2)), dim3(NT), 0, smx_s(stream), a);
        else hipLaunchKernelGGL((lstm_fwdk_kernel<7, 1, false>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && !quad && H <= 128) {
        if (B >= 512) hipLaunchKernelGGL((lstm_fwdk_kernel<8, 2, false>), dim3((unsigned)((B + 1) / 2)), dim3(NT), 0, smx_s(stream), a);
        else hipLaunchKernelGGL((lstm_fwdk_kernel<8, 1, false>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && H <= 100) {
        hipLaunchKernelGGL((lstm_fwdq_kernel<25>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && H <= 128) {
        hipLaunchKernelGGL((lstm_fwdq_kernel<32>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && H <= 100) {
        hipLaunchKernelGGL((lstm_fwd1_kernel<25>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && H <= 128) {
        hipLaunchKernelGGL((lstm_fwd1_kernel<32>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (H <= 100) {
        hipLaunchKernelGGL((lstm_fwd4_kernel<25>), dim3((unsigned)((B + RB4 - 1) / RB4)), dim3(NT),
                           lds4_fwd(25), smx_s(stream), a);
    } else if (H <= 4 * KQ4) {
        hipLaunchKernelGGL((lstm_fwd4_kernel<KQ4>), dim3((unsigned)((B + RB4 - 1) / RB4)), dim3(NT),
                           lds4_fwd(KQ4), smx_s(stream), a);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void*>(&lstm_fwd_kernel<KG_MAX>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(KG_MAX, 16 * KG_MAX));
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL((lstm_fwd_kernel<KG_MAX>), dim3(blocks), dim3(NT),
                           lds_bytes(KG_MAX, H), smx_s(stream), a);
    }
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_lstm_backward_f32(const smx_lstm_t* net, const float* x, int64_t B, int32_t T,
                                     const float* c0, const float* gates, const float* cs,
                                     const float* hprev, const float* dout, float* dgates,
                                     float* grads, const int32_t* stop_flag, float* ws,
                                     int64_t ws_floats, smx_stream_t stream) {
    SMX_REQUIRE(net && x && gates && cs && hprev && dout && dgates && grads, SMX_E_NULL);
    SMX_REQUIRE(net->W_hh, SMX_E_NULL);
    const int H = net->H, D = net->D;
    SMX_REQUIRE(B > 0 && T > 0 && H > 0 && D > 0 && B * T * 4 * H < (1ll << 31), SMX_E_SHAPE);
    SMX_REQUIRE(H % 4 == 0 && H <= 16 * KG_MAX, SMX_E_UNSUPPORTED);
    BwdArgs a;
    a.W_hh = net->W_hh; a.c0 = c0; a.gates = gates; a.cs = cs; a.dout = dout; a.dgates = dgates;
    a.stop = stop_flag; a.B = (int)B; a.T = T; a.H = H;
    const int blocks = (int)((B + RB - 1) / RB);
    static const bool mfma4 = getenv("SMX_LSTM_MFMA4") != nullptr;
    static const bool v1 = getenv("SMX_LSTM_V1") != nullptr;
    static const bool quad = getenv("SMX_LSTM_QUAD") != nullptr;
    static const bool no_mrows = getenv("SMX_LSTM_NO_MROWS") != nullptr;
    static const long mrows_min = getenv("SMX_LSTM_MROWS_MIN") ? atol(getenv("SMX_LSTM_MROWS_MIN")) : 512;
    if (!no_mrows && !mfma4 && !v1 && !quad && H <= 112 && B >= mrows_min) {
        const dim3 grid((unsigned)((B + 3) / 4));
        if (H <= 100) hipLaunchKernelGGL((lstm_bwdm_kernel<25>), grid, dim3(NT), 0, smx_s(stream), a);
        else hipLaunchKernelGGL((lstm_bwdm_kernel<28>), grid, dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && !quad && H <= 104) {
        if (B >= 512) hipLaunchKernelGGL((lstm_bwdk_kernel<13, 2>), dim3((unsigned)((B + 1) / 2)), dim3(NT), 0, smx_s(stream), a);
        else hipLaunchKernelGGL((lstm_bwdk_kernel<13, 1>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && !quad && H <= 128) {
        if (B >= 512) hipLaunchKernelGGL((lstm_bwdk_kernel<16, 2>), dim3((unsigned)((B + 1) / 2)), dim3(NT), 0, smx_s(stream), a);
        else hipLaunchKernelGGL((lstm_bwdk_kernel<16, 1>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && H <= 100) {
        hipLaunchKernelGGL((lstm_bwdq_kernel<25>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && !v1 && H <= 128) {
        hipLaunchKernelGGL((lstm_bwdq_kernel<32>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && H <= 100) {
        hipLaunchKernelGGL((lstm_bwd1_kernel<25>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (!mfma4 && H <= 128) {
        hipLaunchKernelGGL((lstm_bwd1_kernel<32>), dim3((unsigned)B), dim3(NT), 0, smx_s(stream), a);
    } else if (H <= 100) {
        hipLaunchKernelGGL((lstm_bwd4_kernel<25>), dim3((unsigned)((B + RB4 - 1) / RB4)), dim3(NT),
                           lds4_bwd(25), smx_s(stream), a);
    } else if (H <= 4 * KQ4) {
        hipLaunchKernelGGL((lstm_bwd4_kernel<KQ4>), dim3((unsigned)((B + RB4 - 1) / RB4)), dim3(NT),
                           lds4_bwd(KQ4), smx_s(stream), a);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void*>(&lstm_bwd_kernel<KG_MAX>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(KG_MAX, 16 * KG_MAX));
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL((lstm_bwd_kernel<KG_MAX>), dim3(blocks), dim3(NT),
                           lds_bytes(KG_MAX, H), smx_s(stream), a);
    }
    SMX_LAUNCH_CHECK();
    // grads = [dW_ih (4H x D) | dW_hh (4H x H) | db_ih (4H) | db_hh (4H)]  (nn.LSTM parameter order)
    float* gWih = grads;
    float* gWhh = gWih + (size_t)4 * H * D;
    float* gbih = gWhh + (size_t)4 * H * H;
    float* gbhh = gbih + 4 * H;
    const int32_t rows = (int32_t)(B * T);
    // both gradients read the same dgates: ONE split-K launch where both are on the 32 x 32 kernel (B*T ~ 10^4 rows; the
    // call falls back to two launches by itself when one of them has a kernel of its own at these shapes)
    return smx_linear_wgrad_splitk_pair_f32(dgates, 4 * H, 4 * H, rows, x, D, gWih, gbih, D, hprev, H, gWhh, gbhh, H, ws,
                                            ws_floats, stream);
}

extern "C" int64_t smx_lstm_backward_ws_floats(int32_t D, int32_t H, int64_t B, int32_t T) {
    const int64_t a = smx_linear_wgrad_ws_floats(4 * H, D, (int32_t)(B * T));
    const int64_t b = smx_linear_wgrad_ws_floats(4 * H, H, (int32_t)(B * T));
    return a + b;             // (both partial sets at once: smx_linear_wgrad_splitk_pair_f32)
}

extern "C" int64_t smx_lstm_param_count(int32_t D, int32_t H) {
    return (int64_t)4 * H * D + (int64_t)4 * H * H + 8 * (int64_t)H;
}
