// micro-benchmark: the dense products of one fused epoch (layer 1: 19 tiles x K 384, layer 2: 13 x 320, dz1: 19 x 224; packed
// weights streamed from L2 as in smx_epoch.hip) in two formulations --
//   A: 16 data rows per workgroup on v_mfma_f32_16x16x4_f32, 128 workgroups (what epoch_fb_kernel does: half the chip)
//   B: 8 data rows per workgroup on v_mfma_f32_4x4x1_16B_f32 (the SAME packed fragments: a lane's float4 is its feature's
//      weights at four k's, the four 16-lane groups work on different k's and are added at the end), 256 workgroups
// Prints microseconds per launch (HIP events over many launches) for both.  hipcc --offload-arch=gfx950 -O3 rows8_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#ifndef EXP
#define EXP 0
#endif
#if EXP & 1
#define MFMA4(a, b, c) ((c) + (a) * (b))
#else
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#endif
constexpr unsigned OOB = 0x80000000u;
constexpr int FNWV = 8, NT = 3;

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld16(rsrc_t R, unsigned off) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(R, off, 0, 0);
    return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
}
template <int R>
struct WFrag { float4 a[NT], b[NT]; float4 x0[R], x1[R]; };

// ROWS16: one data word pair per chunk (row = lane & 15); else R row groups of 4 rows (row = 4 rg + (lane & 3))
template <int R>
__device__ __forceinline__ void ld_wfrag(WFrag<R>& f, rsrc_t rw, const unsigned (&wo)[NT], const float* bp, int ldi, int c) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const unsigned o = wo[g] + (unsigned)c * 2048u;
        f.a[g] = ld16(rw, o);
        f.b[g] = ld16(rw, o + 1024u);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#if EXP & 2
        f.x0[r] = make_float4(1.f, 2.f, 3.f, 4.f); f.x1[r] = f.x0[r];
#else
        f.x0[r] = *(const float4*)(bp + 4 * r * ldi + 32 * c);
        f.x1[r] = *(const float4*)(bp + 4 * r * ldi + 32 * c + 4);
#endif
    }
}
__device__ __forceinline__ void mma16(f32x4 (&acc)[NT], const WFrag<1>& f) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.a[g].x, f.x0[0].x, acc[g]); acc[g] = MFMA16(f.a[g].y, f.x0[0].y, acc[g]);
        acc[g] = MFMA16(f.a[g].z, f.x0[0].z, acc[g]); acc[g] = MFMA16(f.a[g].w, f.x0[0].w, acc[g]);
    }
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.b[g].x, f.x1[0].x, acc[g]); acc[g] = MFMA16(f.b[g].y, f.x1[0].y, acc[g]);
        acc[g] = MFMA16(f.b[g].z, f.x1[0].z, acc[g]); acc[g] = MFMA16(f.b[g].w, f.x1[0].w, acc[g]);
    }
}
// A operand = the data rows (lane & 3 = row of the group, the same in every block), B operand = the lane's weight
template <int R>
__device__ __forceinline__ void mma4(f32x4 (&acc)[NT][R], const WFrag<R>& f) {
    // k step outermost: consecutive instructions go to DIFFERENT accumulators (a dependent 4x4x1 waits for the previous pass)
#define STEP(X, W, E)                                                        \
    _Pragma("unroll") for (int g = 0; g < NT; ++g)                           \
        _Pragma("unroll") for (int r = 0; r < R; ++r) acc[g][r] = MFMA4(f.X[r].E, f.W[g].E, acc[g][r]);
    STEP(x0, a, x) STEP(x0, a, y) STEP(x0, a, z) STEP(x0, a, w)
    STEP(x1, b, x) STEP(x1, b, y) STEP(x1, b, z) STEP(x1, b, w)
#undef STEP
}

struct Layer { const float* W; int tiles, C2; };
struct Args { Layer l[3]; float* out; int steps; };

template <int RG>
__global__ __launch_bounds__(512) void k_chain(Args A) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ldi = 420;
    for (int i = tid; i < 16 * ldi; i += 512) sm[i] = 0.001f * (i & 255);
    __syncthreads();
    float s = 0.f;
#pragma unroll 1
  for (int step = 0; step < A.steps; ++step) {
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        const Layer L = A.l[l];
        const rsrc_t rw = make_rsrc(L.W, (unsigned)L.tiles * (unsigned)L.C2 * 2048u);
#pragma unroll 1
        for (int tb = 0; tb < L.tiles; tb += FNWV * NT) {
            unsigned wo[NT];
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const int t = tb + wv + FNWV * g;
                wo[g] = t < L.tiles ? ((unsigned)t * (unsigned)L.C2 * 512u + (unsigned)lane * 4u) * 4u : OOB;
            }
            if (tb + wv >= L.tiles) continue;
            if (RG == 0) {
                const float* bp = sm + (lane & 15) * ldi + 8 * (lane >> 4);
                f32x4 acc[NT];
#pragma unroll
                for (int g = 0; g < NT; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                WFrag<1> P0, P1, Q0, Q1;
                auto cl = [&](int c) { return c < L.C2 ? c : L.C2 - 1; };
                ld_wfrag<1>(P0, rw, wo, bp, ldi, 0);
                __builtin_amdgcn_sched_barrier(0);
                ld_wfrag<1>(P1, rw, wo, bp, ldi, 1);
                int c = 0;
#pragma unroll 1
                for (; c + 4 <= L.C2; c += 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<1>(Q0, rw, wo, bp, ldi, c + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mma16(acc, P0);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<1>(Q1, rw, wo, bp, ldi, c + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    mma16(acc, P1);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<1>(P0, rw, wo, bp, ldi, cl(c + 4));
                    __builtin_amdgcn_sched_barrier(0);
                    mma16(acc, Q0);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<1>(P1, rw, wo, bp, ldi, cl(c + 5));
                    __builtin_amdgcn_sched_barrier(0);
                    mma16(acc, Q1);
                }
                if (c < L.C2) {
                    __builtin_amdgcn_sched_barrier(0);
                    mma16(acc, P0);
                    mma16(acc, P1);
                }
                for (int g = 0; g < NT; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
            } else {
                const float* bp = sm + (lane & 3) * ldi + 8 * (lane >> 4);
                constexpr int R = RG ? RG : 1;
                f32x4 acc[NT][R];
#pragma unroll
                for (int g = 0; g < NT; ++g) for (int r = 0; r < R; ++r) acc[g][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                WFrag<R> P0, P1, Q0, Q1;
                auto cl = [&](int c) { return c < L.C2 ? c : L.C2 - 1; };
                ld_wfrag<R>(P0, rw, wo, bp, ldi, 0);
                __builtin_amdgcn_sched_barrier(0);
                ld_wfrag<R>(P1, rw, wo, bp, ldi, 1);
                int c = 0;
#pragma unroll 1
                for (; c + 4 <= L.C2; c += 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<R>(Q0, rw, wo, bp, ldi, c + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mma4<R>(acc, P0);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<R>(Q1, rw, wo, bp, ldi, c + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    mma4<R>(acc, P1);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<R>(P0, rw, wo, bp, ldi, cl(c + 4));
                    __builtin_amdgcn_sched_barrier(0);
                    mma4<R>(acc, Q0);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_wfrag<R>(P1, rw, wo, bp, ldi, cl(c + 5));
                    __builtin_amdgcn_sched_barrier(0);
                    mma4<R>(acc, Q1);
                }
                if (c < L.C2) {
                    __builtin_amdgcn_sched_barrier(0);
                    mma4<R>(acc, P0);
                    mma4<R>(acc, P1);
                }
                for (int g = 0; g < NT; ++g)
                    for (int r = 0; r < R; ++r) {
                        f32x4 v = acc[g][r];
                        for (int k = 0; k < 4; ++k) {          // the four k groups meet
                            float x = v[k];
                            x += __shfl_xor(x, 16, 64);
                            x += __shfl_xor(x, 32, 64);
                            s += x;
                        }
                    }
            }
        }
        __syncthreads();
    }
  }
    A.out[blockIdx.x * 512 + tid] = s;
}

template <int RG>
double run(Args A, int wgs, int steps) {
    hipFuncSetAttribute((const void*)k_chain<RG>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int lds = 100 * 1024;
    A.steps = steps;
    hipLaunchKernelGGL((k_chain<RG>), dim3(wgs), dim3(512), lds, 0, A);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_chain<RG>), dim3(wgs), dim3(512), lds, 0, A);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / steps;
}

int main() {
    float *W, *out;
    hipMalloc(&W, 8 << 20); hipMalloc(&out, 1 << 22);
    hipMemset(W, 0, 8 << 20);
    Args A;
    // the policy's three layers at D = 376, [300, 200], A = 17: what one step of the rollout kernel multiplies
    A.l[0] = {W, 19, 12};
    A.l[1] = {W + (1 << 19), 13, 10};
    A.l[2] = {W + (1 << 20), 2, 8};
    A.out = out;
    const int steps = 128;
    printf("per step of the three policy layers, 1024 actors (weights streamed from L2 by every workgroup):\n");
    printf("16 rows x  64 workgroups (16x16x4): %.2f us\n", run<0>(A, 64, steps));
    printf(" 8 rows x 128 workgroups (4x4x1):   %.2f us\n", run<2>(A, 128, steps));
    printf(" 4 rows x 256 workgroups (4x4x1):   %.2f us\n", run<1>(A, 256, steps));
    printf(" 4 rows x 128 workgroups (4x4x1):   %.2f us (512 actors)\n", run<1>(A, 128, steps));
    printf(" 8 rows x 256 workgroups (4x4x1):   %.2f us (2048 actors)\n", run<2>(A, 256, steps));
    printf("16 rows x 256 workgroups (16x16x4): %.2f us (4096 actors)\n", run<0>(A, 256, steps));
    // the same number of tile-chunks as ONE balanced layer (24 tiles x 16 chunks = 384: three tiles per wave, one ramp) and
    // as three balanced layers of 8 tiles x 16 chunks
    Args B1 = A;
    B1.l[0] = {W, 24, 16}; B1.l[1] = {W, 0, 2}; B1.l[2] = {W, 0, 2};
    printf(" 4 rows x 256 workgroups, ONE balanced layer of 24 tiles x 16 chunks: %.2f us\n", run<1>(B1, 256, steps));
    Args B3 = A;
    B3.l[0] = {W, 8, 16}; B3.l[1] = {W + (1 << 19), 8, 16}; B3.l[2] = {W + (1 << 20), 8, 16};
    printf(" 4 rows x 256 workgroups, THREE balanced layers of 8 tiles x 16 chunks: %.2f us\n", run<1>(B3, 256, steps));
    Args B4 = A;
    B4.l[0] = {W, 24, 6}; B4.l[1] = {W + (1 << 19), 24, 6}; B4.l[2] = {W + (1 << 20), 16, 6};
    printf(" 4 rows x 256 workgroups, three layers of 24/24/16 tiles x 6 chunks (384 tile-chunks): %.2f us\n", run<1>(B4, 256, steps));
    return 0;
}
