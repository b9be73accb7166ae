// micro-benchmark: how fast ONE CU pulls a packed weight set that sits in L2 (every workgroup of the grid streams the same
// 744 KB, as the row-block kernels do), as a function of the waves per workgroup, the 1 KB wave-loads each keeps in flight and
// the cache-policy bits of the load.  No MFMA: the loaded words are XORed into a register.  Prints bytes per clock and CU.
//   hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int INFL, int AUX>
__global__ void k_stream(const float* W, unsigned bytes, int reps, unsigned* out) {
    const rsrc_t R = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    const unsigned kb = bytes >> 10;                  // 1 KB pieces; wave w takes pieces w, w + nw, ...
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (unsigned p = wv; p < kb; p += nw * INFL) {
            u32x4 v[INFL];
#pragma unroll
            for (int i = 0; i < INFL; ++i) {
                const unsigned q = p + i * nw;
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(R, q < kb ? q * 1024u + lane * 16u : 0x80000000u, 0, AUX);
            }
#pragma unroll
            for (int i = 0; i < INFL; ++i) acc ^= v[i];
        }
    }
    out[blockIdx.x * blockDim.x + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int INFL, int AUX>
void run(const float* W, unsigned bytes, unsigned* out, int wgs, int threads, const char* what) {
    const int reps = 64;
    hipLaunchKernelGGL((k_stream<INFL, AUX>), dim3(wgs), dim3(threads), 0, 0, W, bytes, 4, out);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_stream<INFL, AUX>), dim3(wgs), dim3(threads), 0, 0, W, bytes, reps, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)bytes * reps / (ms * 1e-3) / 2.4e9;
    printf("%-44s %3d workgroups x %4d threads, %2d loads in flight per wave: %6.1f B/clk/CU (2.4 GHz)  %7.2f TB/s chip\n",
           what, wgs, threads, INFL, per_cu, (double)bytes * reps * wgs / (ms * 1e-3) / 1e12);
}

int main() {
    const unsigned bytes = 744 * 1024;
    float* W; unsigned* out;
    (void)hipMalloc(&W, 8 << 20); (void)hipMalloc(&out, 1 << 22);
    (void)hipMemset(W, 0, 8 << 20);
    for (int wgs : {256, 128, 32, 8, 1}) {
        run<6, 0>(W, bytes, out, wgs, 512, "aux 0");
    }
    run<2, 0>(W, bytes, out, 256, 512, "aux 0");
    run<12, 0>(W, bytes, out, 256, 512, "aux 0");
    run<6, 0>(W, bytes, out, 256, 256, "aux 0");
    run<6, 0>(W, bytes, out, 256, 1024, "aux 0");
    run<12, 0>(W, bytes, out, 256, 1024, "aux 0");
    run<6, 1>(W, bytes, out, 256, 512, "aux 1 (sc0)");
    run<6, 2>(W, bytes, out, 256, 512, "aux 2 (nt)");
    run<6, 3>(W, bytes, out, 256, 512, "aux 3 (sc0 nt)");
    run<6, 16>(W, bytes, out, 256, 512, "aux 16 (sc1)");
    run<6, 17>(W, bytes, out, 256, 512, "aux 17 (sc0 sc1)");
    run<6, 18>(W, bytes, out, 256, 512, "aux 18 (sc1 nt)");
    return 0;
}
