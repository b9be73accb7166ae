// micro-benchmark of the forward inner loop of smx_epoch.hip (5 tiles x 12 chunks per wave, 4 waves per
// workgroup, 128 workgroups): cycles per 32-wide K chunk under variations of the operand path
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
constexpr unsigned OOB = 0x80000000u;
constexpr int NT = 5;

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld16(rsrc_t R, unsigned off) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(R, off, 0, 0);
    return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
}
struct WFrag { float4 a[NT], b[NT]; float4 x0, x1; };

template <int MODE>   // 0 row-major (uncoalesced), 1 packed (coalesced), 2 no global loads, 3 global plain pointer loads row-major
__device__ __forceinline__ void ld_wfrag(WFrag& f, rsrc_t rw, const float* W, const unsigned (&wo)[NT], const float* bp, int c, int nch) {
    const unsigned cs = MODE == 1 ? 2048u : 128u, hs = MODE == 1 ? 1024u : 16u;
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        if (MODE == 2) { f.a[g] = make_float4(1, 2, 3, 4); f.b[g] = make_float4(1, 2, 3, 4); continue; }
        const unsigned o = (c < nch) ? wo[g] + (unsigned)c * cs : OOB;
        if (MODE == 3) {
            const unsigned oo = (c < nch && o < OOB) ? o : 0;
            f.a[g] = *(const float4*)((const char*)W + oo);
            f.b[g] = *(const float4*)((const char*)W + oo + hs);
        } else {
            f.a[g] = ld16(rw, o);
            f.b[g] = ld16(rw, o + hs);
        }
    }
    const float* q = bp + 32 * (c < nch ? c : 0);
    f.x0 = *(const float4*)(q);
    f.x1 = *(const float4*)(q + 4);
}
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[NT], const WFrag& f) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.a[g].x, f.x0.x, acc[g]); acc[g] = MFMA16(f.a[g].y, f.x0.y, acc[g]);
        acc[g] = MFMA16(f.a[g].z, f.x0.z, acc[g]); acc[g] = MFMA16(f.a[g].w, f.x0.w, acc[g]);
    }
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        acc[g] = MFMA16(f.b[g].x, f.x1.x, acc[g]); acc[g] = MFMA16(f.b[g].y, f.x1.y, acc[g]);
        acc[g] = MFMA16(f.b[g].z, f.x1.z, acc[g]); acc[g] = MFMA16(f.b[g].w, f.x1.w, acc[g]);
    }
}

template <int MODE, bool COND>
__global__ __launch_bounds__(256) void k_loop(const float* W, int M, int K, float* out, long long* cyc, int reps) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fm = lane & 15, kq = lane >> 4;
    const int ldi = ((K + 31) & ~31) + 4;
    for (int i = tid; i < 16 * ldi; i += 256) sm[i] = 0.001f * i;
    __syncthreads();
    const int nch = (K + 31) >> 5;
    const rsrc_t rw = make_rsrc(W, (unsigned)M * K * 4u);
    unsigned wo[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const int t = wv + 4 * g, row = 16 * t + fm;
        const unsigned o = MODE == 1 ? ((unsigned)t * nch * 512u + (unsigned)(kq * 16 + fm) * 4u) * 4u
                                     : ((unsigned)row * K + 8u * kq) * 4u;
        wo[g] = row < M ? o : OOB;
    }
    const float* bp = sm + fm * ldi + 8 * kq;
    f32x4 acc[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        WFrag P, Q;
        ld_wfrag<MODE>(P, rw, W, wo, bp, 0, nch);
#pragma unroll 1
        for (int c = 0; c < nch; c += 2) {
            __builtin_amdgcn_sched_barrier(0);
            ld_wfrag<MODE>(Q, rw, W, wo, bp, c + 1, nch);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(acc, P);
            __builtin_amdgcn_sched_barrier(0);
            ld_wfrag<MODE>(P, rw, W, wo, bp, c + 2, nch);
            __builtin_amdgcn_sched_barrier(0);
            if (!COND || c + 1 < nch) mma_chunk(acc, Q);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int g = 0; g < NT; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int MODE, bool COND>
void run(const char* name, const float* W, float* out, long long* cyc) {
    const int M = 300, K = 384, wgs = 128, reps = 4;
    long long h[1024];
    const int lds = 16 * (K + 4) * 4 + 70000;
    hipFuncSetAttribute((const void*)k_loop<MODE, COND>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_loop<MODE, COND>), dim3(wgs), dim3(256), lds, 0, W, M, K, out, cyc, reps);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (int i = 0; i < wgs * 4; ++i) { s += h[i]; if (h[i] > mx) mx = h[i]; }
    printf("%-44s %7.0f cycles per chunk (mean), max %7.0f   [floor 1280]\n", name, s / (wgs * 4) / (reps * 12.0), mx / (reps * 12.0));
}

int main() {
    float *W, *out; long long* cyc;
    hipMalloc(&W, 4 << 20); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 20);
    hipMemset(W, 0, 4 << 20);
    run<2, false>("no global loads, even chunk count", W, out, cyc);
    run<2, true>("no global loads, conditional 2nd mma", W, out, cyc);
    run<1, false>("packed (coalesced) buffer loads", W, out, cyc);
    run<1, true>("packed, conditional 2nd mma", W, out, cyc);
    run<0, false>("row-major buffer loads", W, out, cyc);
    run<0, true>("row-major, conditional 2nd mma", W, out, cyc);
    run<3, true>("row-major plain global loads, conditional", W, out, cyc);
    return 0;
}
