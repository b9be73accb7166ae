// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 per wave with 1..8 waves per workgroup,
// with and without interleaved buffer loads / LDS reads (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NACC>
__global__ void k_mfma(float* out, long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int g = 0; g < NACC; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int g = 0; g < NACC; ++g) acc[g] = MFMA16(a, b, acc[g]);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int g = 0; g < NACC; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 20);
    long long h[8192];
    const int iters = 50;
    for (int nw = 1; nw <= 8; nw *= 2) {
        for (int wgs : {64, 128, 256}) {
            hipLaunchKernelGGL(k_mfma<5>, dim3(wgs), dim3(64 * nw), 0, 0, out, cyc, iters);
            hipLaunchKernelGGL(k_mfma<5>, dim3(wgs), dim3(64 * nw), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, wgs * nw * 8, hipMemcpyDeviceToHost);
            double s = 0, mx = 0;
            for (int i = 0; i < wgs * nw; ++i) { s += h[i]; if (h[i] > mx) mx = h[i]; }
            printf("waves/wg %d wgs %3d: %.1f cycles per MFMA per wave (mean), max %.1f\n", nw, wgs,
                   s / (wgs * nw) / (iters * 40.0), mx / (iters * 40.0));
        }
    }
    return 0;
}
