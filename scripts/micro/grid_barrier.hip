// micro-benchmark: cost of a device-wide barrier (atomic ticket + spin) vs kernel launches
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void barrier_kernel(unsigned* counter, int nbar, float* out) {
    const unsigned nblk = gridDim.x;
    float v = threadIdx.x;
    for (int b = 0; b < nbar; ++b) {
        v = v * 1.0001f + 1.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)(b + 1) * nblk;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (v == -1.f) out[0] = v;
}

__global__ void empty_kernel(float* out, int k) {
    if (k == -1) out[0] = 1.f;
}

int main() {
    unsigned* counter; float* out;
    hipMalloc(&counter, 4); hipMalloc(&out, 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {64, 256, 512, 640}) {
        for (int nbar : {1, 101}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemsetAsync(counter, 0, 4, st);
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(barrier_kernel, dim3(blocks), dim3(256), 0, st, counter, nbar, out);
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("blocks %4d barriers %4d : %.2f us total\n", blocks, nbar, best * 1e3);
        }
    }
    // dependent launches
    for (int n : {1, 101}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, st);
            for (int k = 0; k < n; ++k) hipLaunchKernelGGL(empty_kernel, dim3(640), dim3(256), 0, st, out, k);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("empty launches %4d : %.2f us total\n", n, best * 1e3);
    }
    return 0;
}
