// Shared device/host helpers for libsurreal_amd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/surreal_amd.h"

#define SMX_WAVE 64

#define SMX_REQUIRE(cond, code) \
    do {                        \
        if (!(cond)) return (code); \
    } while (0)

#define SMX_LAUNCH_CHECK()                              \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return (int)e__;         \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float smx_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Deterministic block sum for blockDim.x <= 1024 (multiple of 64). `red` holds >= 16 floats.
static __device__ __forceinline__ float smx_block_sum(float v, float* red) {
    v = smx_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];  // fixed order -> same value in every thread
    return t;
}

static __device__ __forceinline__ double smx_wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

static inline hipStream_t smx_s(smx_stream_t s) { return (hipStream_t)s; }

// compute units of the current device (what bounds the workgroups a launch with in-launch waits may have); 64 if the
// runtime cannot say
static inline int smx_cu_count() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            n_cu = n;
        else
            n_cu = 64;
    }
    return n_cu;
}
