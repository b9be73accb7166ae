// The 4-row MFMA loop (round 6): a workgroup carries 4 RG data rows (RG row groups of four) through a dense layer in
// TRANSPOSED form on v_mfma_f32_4x4x1_16B_f32, with the SAME fragment-order packed weights as the 16-row loop of
// smx_epoch_mma.inc.h (smx_epoch_pack.inc.h: lane l = 16 kq + i of (tile t, chunk c, half h) holds
// W[16 t + i][32 c + 8 kq + 4 h + 0..3]).
//
// One v_mfma_f32_4x4x1 is 16 independent 4 x 4 outer products: lane 4b + i supplies A_b[i], lane 4b + j supplies B_b[j] and
// holds column j of the result (VGPR r = row r).  With A = the data rows (lane l: row l & 3 of the group, at the k of ITS
// 16-lane group kq = l >> 4) and B = the lane's own weight word, block b = 4 kq + (i >> 2) multiplies four features by four
// rows at one k: an instruction covers 16 features x 4 rows x 4 k's (one per kq) at the nominal FP32 rate (512 FLOP, 8
// cycles).  A lane ends with the partial sums of ITS feature over the k's of its kq for the four rows; the four kq groups
// (lanes l, l ^ 16, l ^ 32, l ^ 48) meet by two cross-lane adds per value after the K loop.
//
// Why: 16-row workgroups put 1024 rows on 64 CUs (one network) -- the rollout kernel ran on a quarter of the chip, bound by
// the matrix pipes of those CUs.  4-row workgroups put 1024 rows on all 256 CUs; per step every workgroup still streams the
// whole packed weight set from L2, which is what then bounds it (scripts/micro/rows4_rollout.hip, the policy's three layers
// at D = 376, [300, 200], 1024 actors: 16 rows x 64 workgroups 19.6 us per step, 8 x 128 14.7, 4 x 256 11.2).
//
// Summation order per (row, feature): k ascending within each of the four kq classes (k mod 32 in [8 kq, 8 kq + 8)), then
// (kq0 + kq1) + (kq2 + kq3) -- NOT the order of the 16-row loop (k ascending in steps of four interleaved over kq by the
// 16x16x4 instruction): results agree with it to fp32 rounding, not bit for bit.
#pragma once

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

template <int NT, int RG>
struct WFrag4 {
    float4 a[NT], b[NT];          // weights: k = 8kq + 0..3 and 8kq + 4..7 of the chunk, per tile
    float4 x0[RG], x1[RG];        // the data rows' words of the same chunk (LDS), per row group
};

template <int NT, int RG>
__device__ __forceinline__ void ld_wfrag4(WFrag4<NT, RG>& f, rsrc_t rw, const unsigned (&wo)[NT], const float* bp, int ldi, int c) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const unsigned o = wo[g] + (unsigned)c * 2048u;     // past the last chunk: past the buffer (0)
        f.a[g] = ld16(rw, o);
        f.b[g] = ld16(rw, o + 1024u);
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        f.x0[r] = *(const float4*)(bp + 4 * r * ldi + 32 * c);
        f.x1[r] = *(const float4*)(bp + 4 * r * ldi + 32 * c + 4);
    }
}

// k step outermost: consecutive instructions go to DIFFERENT accumulators (a dependent 4x4x1 waits for the previous one's pass)
template <int NT, int RG>
__device__ __forceinline__ void mma4_chunk(f32x4 (&acc)[NT][RG], const WFrag4<NT, RG>& f) {
#define SMX_STEP4(X, W, E)                                                   \
    _Pragma("unroll") for (int g = 0; g < NT; ++g)                           \
        _Pragma("unroll") for (int r = 0; r < RG; ++r) acc[g][r] = MFMA4(f.X[r].E, f.W[g].E, acc[g][r]);
    SMX_STEP4(x0, a, x) SMX_STEP4(x0, a, y) SMX_STEP4(x0, a, z) SMX_STEP4(x0, a, w)
    SMX_STEP4(x1, b, x) SMX_STEP4(x1, b, y) SMX_STEP4(x1, b, z) SMX_STEP4(x1, b, w)
#undef SMX_STEP4
}

// tiles t0, t0 + tstep, ... (NT of them; tiles >= `tiles` are out-of-range operands) over the C2 (even) K chunks; the data
// rows at in_lds[row * ldi + k].  Four register stages as in fwd_tiles (smx_epoch_mma.inc.h); loads past the last chunk are
// past the buffer (zeros) on the weight side and read LDS words that exist (the caller's tiles are padded to C2 chunks + 2).
template <int NT, int RG, bool SHALLOW = false>
__device__ __forceinline__ void fwd_tiles4(f32x4 (&acc)[NT][RG], rsrc_t rw, int tiles, int C2, const float* in_lds,
                                           int ldi, int t0, int tstep, int lane) {
    unsigned wo[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const int t = t0 + tstep * g;
        wo[g] = (t < tiles) ? ((unsigned)t * (unsigned)C2 * 512u + (unsigned)lane * 4u) * 4u : OOB;
    }
    const float* bp = in_lds + (lane & 3) * ldi + 8 * (lane >> 4);
    auto cl = [&](int c) { return c < C2 ? c : C2 - 1; };      // (the LDS side of a prefetch past the end: a valid chunk)
    // SHALLOW: ONE chunk in flight behind the one being multiplied (two register stages) instead of two (four).  Which is
    // faster depends on how many workgroups share the L2s: with all 256 CUs streaming (the rollout kernel) the shallow queue
    // wins by 3 % (1.59 -> 1.55 ms at 1024 actors; THREE in flight: 1.63), with 128 (the DDPG chains) the deeper one by 5 %.
    if (SHALLOW) {
        WFrag4<NT, RG> P0, P1;
        ld_wfrag4<NT, RG>(P0, rw, wo, bp, ldi, 0);
        int c = 0;
#pragma unroll 1
        for (; c + 2 <= C2; c += 2) {
            __builtin_amdgcn_sched_barrier(0);
            ld_wfrag4<NT, RG>(P1, rw, wo, bp, ldi, c + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma4_chunk<NT, RG>(acc, P0);
            __builtin_amdgcn_sched_barrier(0);
            ld_wfrag4<NT, RG>(P0, rw, wo, bp, ldi, cl(c + 2));
            __builtin_amdgcn_sched_barrier(0);
            mma4_chunk<NT, RG>(acc, P1);
        }
        return;
    }
    WFrag4<NT, RG> P0, P1, Q0, Q1;
    ld_wfrag4<NT, RG>(P0, rw, wo, bp, ldi, 0);
    __builtin_amdgcn_sched_barrier(0);
    ld_wfrag4<NT, RG>(P1, rw, wo, bp, ldi, 1);
    int c = 0;
#pragma unroll 1
    for (; c + 4 <= C2; c += 4) {
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag4<NT, RG>(Q0, rw, wo, bp, ldi, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma4_chunk<NT, RG>(acc, P0);
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag4<NT, RG>(Q1, rw, wo, bp, ldi, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma4_chunk<NT, RG>(acc, P1);
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag4<NT, RG>(P0, rw, wo, bp, ldi, cl(c + 4));
        __builtin_amdgcn_sched_barrier(0);
        mma4_chunk<NT, RG>(acc, Q0);
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag4<NT, RG>(P1, rw, wo, bp, ldi, cl(c + 5));
        __builtin_amdgcn_sched_barrier(0);
        mma4_chunk<NT, RG>(acc, Q1);
    }
    if (c < C2) {                                      // C2 = 4j + 2: the last two chunks are in flight
        __builtin_amdgcn_sched_barrier(0);
        mma4_chunk<NT, RG>(acc, P0);
        mma4_chunk<NT, RG>(acc, P1);
    }
}

// the four kq groups meet: afterwards every lane of a feature holds the full sums of its four rows.  Two register-file
// swaps and two adds per value (v_permlane16_swap: odd 16-lane rows of one copy against even rows of the other ->
// x0 + x1 | x2 + x3 in every row pair; v_permlane32_swap: the halves) -- as __shfl_xor (ds_bpermute, an LDS round trip
// per hop) the 48 dependent pairs of a layer's epilogue cost more than the products they followed (measured in
// epoch_fb8_kernel: layer 1's epilogue 7.3 k cycles against 4.4 k on 16-row blocks).
__device__ __forceinline__ float meet_kq1(float x) {
    unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    u = __float_as_uint(x);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ f32x4 meet_kq(f32x4 v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = meet_kq1(v[r]);
    return v;
}

// the same meeting when lane (fm, kq) only needs ROW kq of its feature (every epilogue: one word per lane): the swaps move
// two registers at a time, so rows 0|1 and 2|3 travel together -- three swaps and three adds instead of eight and eight
// and a four-way select.  Same additions in the same order as meet_kq ((kq0 + kq1) + (kq2 + kq3)): bit-identical.
__device__ __forceinline__ float meet_rows(f32x4 v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
    // lane rows 0 | 1 | 2 | 3 now hold: row 0 over kq 0,1 | row 1 over kq 0,1 | row 0 over kq 2,3 | row 1 over kq 2,3 (a), rows 2, 3 (b)
    const float u = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const float t = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(t), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}
