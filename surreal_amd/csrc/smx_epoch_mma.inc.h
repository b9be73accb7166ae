// The row-block MFMA loop shared by the fused epoch kernels (smx_epoch.hip) and the persistent rollout kernel
// (smx_rollout.hip): a workgroup of four wavefronts carries 16 data rows through a dense layer in TRANSPOSED form
// (h^T = W . x^T on v_mfma_f32_16x16x4_f32), weights streamed from the fragment-order packed copy
// (smx_epoch_pack.inc.h).  Included inside an anonymous namespace; both users perform the same operations in the
// same order, so their results are bit-identical.
#pragma once

// workgroup barrier for data exchanged through LDS: does NOT wait for the wave's global stores
// (a user that also includes smx_ppo_loss.inc.h defines it BEFORE that header, which otherwise falls back to __syncthreads())
#ifndef SMX_LDS_BARRIER
#define SMX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

constexpr int ER = 16;            // data rows per workgroup (= MFMA N)
constexpr int NWV = 4;            // waves per workgroup, one per SIMD
constexpr int NTH = 64 * NWV;
constexpr int TG = 5;             // feature tiles a wave carries per pass, forward (4 accumulator VGPRs each)
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

// ---------------------------------------------------------------------------------------------
// forward: accT[g] (16 features x 16 rows) += W[16 t_g .. +16, :K] . inT[:K, 16 rows]
// lane l: fm = l & 15 is the feature within the tile for the A operand and the data row for the
// B operand and the C fragment; kq = l >> 4.  A 32-wide K chunk c is eight MFMA steps (h, r),
// h = 0/1, r = 0..3, step (h, r) multiplies k = 32c + 8kq + 4h + r: each lane fetches 32 contiguous
// bytes of its weight row and two 16-byte words of its data row per chunk.  Weight rows >= M and
// bytes past the matrix are out-of-range buffer loads (0); bytes past K inside the matrix belong to
// the next row and meet the zero padding of the data tile.
//
// CODE SIZE is a first-order cost here: a workgroup runs each instruction stream once or a few
// times, so every kilobyte of unrolled code is an instruction-cache miss chain (the first version --
// one inlined copy per layer and per tile count, 56 KB -- spent 10 of its 31 us fetching code).  The
// three layers therefore share ONE loop body, and the tile count per wave picks among few variants
// (a missing tile is an out-of-range operand, its MFMAs run on zeros).
// ---------------------------------------------------------------------------------------------
template <int NT>
struct WFrag {
    float4 a[NT], b[NT];      // weights: k = 8kq + 0..3 and 8kq + 4..7 of the chunk, per tile
    float4 x0, x1;            // the data row's words of the same chunk (LDS)
};

// Packed weights (smx_epoch_pack_f32): [tile][chunk][half][lane][4 floats], zero padded to whole tiles
// and an EVEN number of chunks.  A load instruction reads one contiguous KB.  (Read from their
// row-major home -- adjacent lanes = adjacent ROWS, 16 cache lines per instruction -- the same loop
// ran at 2500 cycles per chunk against 1280 of MFMA issue; packed: 1540.  scripts/micro/fwd_loop.hip)
template <int NT>
__device__ __forceinline__ void ld_wfrag(WFrag<NT>& f, rsrc_t rw, const unsigned (&wo)[TG], const float* bp, int c) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const unsigned o = wo[g] + (unsigned)c * 2048u;     // past the last chunk: past the buffer (0)
        f.a[g] = ld16(rw, o);
        f.b[g] = ld16(rw, o + 1024u);
    }
    f.x0 = *(const float4*)(bp + 32 * c);
    f.x1 = *(const float4*)(bp + 32 * c + 4);
}

template <int NT>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[TG], const WFrag<NT>& f) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.a[g].x, f.x0.x, acc[g]);
        acc[g] = MFMA16(f.a[g].y, f.x0.y, acc[g]);
        acc[g] = MFMA16(f.a[g].z, f.x0.z, acc[g]);
        acc[g] = MFMA16(f.a[g].w, f.x0.w, acc[g]);
    }
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.b[g].x, f.x1.x, acc[g]);
        acc[g] = MFMA16(f.b[g].y, f.x1.y, acc[g]);
        acc[g] = MFMA16(f.b[g].z, f.x1.z, acc[g]);
        acc[g] = MFMA16(f.b[g].w, f.x1.w, acc[g]);
    }
}

// tiles t0, t0 + tstep, ... (NT of them; tiles >= `tiles` are out-of-range operands) over the C2 (even)
// K chunks.  Four register stages: while the MFMAs of one chunk issue (NT x 8 x 32 cycles), the next
// two chunks' 2 NT weight loads and 2 LDS reads each are in flight.  The scheduling barriers pin that order --
// left alone, hipcc sinks each load to just in front of the MFMA that consumes it and waits for L2
// there.  Both halves of the loop body are unconditional: a branch around the MFMAs makes hipcc
// park the accumulators in VGPRs and copy all of them to the MFMA registers and back every chunk.
template <int NT>
__device__ __forceinline__ void fwd_tiles(f32x4 (&acc)[TG], rsrc_t rw, int tiles, int C2, const float* in_lds,
                                          int ldi, int t0, int tstep, int lane) {
    unsigned wo[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g) {
        const int t = t0 + tstep * g;
        wo[g] = (g < NT && t < tiles) ? ((unsigned)t * (unsigned)C2 * 512u + (unsigned)lane * 4u) * 4u : OOB;
    }
    const float* bp = in_lds + (lane & 15) * ldi + 8 * (lane >> 4);
    // prefetch distance TWO chunks (the packed weights were just rewritten on other XCDs: a first touch
    // is a ~1 us trip to the memory-side cache, longer than one chunk's 1280 cycles of MFMA issue)
    WFrag<NT> P0, P1, Q0, Q1;
    ld_wfrag<NT>(P0, rw, wo, bp, 0);
    __builtin_amdgcn_sched_barrier(0);
    ld_wfrag<NT>(P1, rw, wo, bp, 1);
    int c = 0;
#pragma unroll 1
    for (; c + 4 <= C2; c += 4) {
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag<NT>(Q0, rw, wo, bp, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk<NT>(acc, P0);
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag<NT>(Q1, rw, wo, bp, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk<NT>(acc, P1);
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag<NT>(P0, rw, wo, bp, c + 4);          // (past the last chunk: past the buffer, zeros)
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk<NT>(acc, Q0);
        __builtin_amdgcn_sched_barrier(0);
        ld_wfrag<NT>(P1, rw, wo, bp, c + 5);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk<NT>(acc, Q1);
    }
    if (c < C2) {                                      // C2 = 4j + 2: the last two chunks are in flight
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk<NT>(acc, P0);
        mma_chunk<NT>(acc, P1);
    }
}


// the same for a K of at most 32 (ONE chunk: the products against the output layer's transposed weights, K = OUT <= 32;
// the packed copy pads to an even number of chunks, the second one is all zeros and is not read): no loop
template <int NT>
__device__ __forceinline__ void fwd_tiles_k32(f32x4 (&acc)[TG], rsrc_t rw, int tiles, const float* in_lds, int ldi,
                                              int t0, int tstep, int lane) {
    unsigned wo[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g) {
        const int t = t0 + tstep * g;
        wo[g] = (g < NT && t < tiles) ? ((unsigned)t * 2u * 512u + (unsigned)lane * 4u) * 4u : OOB;
    }
    const float* bp = in_lds + (lane & 15) * ldi + 8 * (lane >> 4);
    WFrag<NT> P0;
    ld_wfrag<NT>(P0, rw, wo, bp, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma_chunk<NT>(acc, P0);
}

