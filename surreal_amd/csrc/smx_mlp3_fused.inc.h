// Shared between the two fused 3-layer forward kernels (smx_mlp3_fused.hip: 32-row wavefronts on
// v_mfma_f32_32x32x2; smx_mlp3_rows16.hip: 16-row wavefronts on v_mfma_f32_16x16x4): the launch
// arguments and the packed weight layout both read.
#pragma once
#include "smx_common.h"

namespace smxf {

struct FusedArgs {
    const float* packed;
    const float* x_main;
    const float* x_tail;
    const float* zmean;
    const float* zstd;
    float* out;
    long total_rows;
    int T0, T1, D, OUT, KC1, out_act, xvec;
    long long* tbuf;   // SMX_FUSED_TIMING builds: per-workgroup phase timestamps (else null)
    int exp;           // SMX_FUSED_TIMING builds: experiment switches (SMX_FUSED_EXP)
    // forward WITH the hidden activations kept for a backward pass (smx_mlp3_forward_rows_f32: the MLPs on top of an
    // LSTM / CNN stem over B*E ~ 10^5 rows): h1 [rows, H1], h2 [rows, H2] row-major, or null
    float* h1_out;
    float* h2_out;
    int H1, H2, out_ld;
    const int* stop;   // device flag (may be null): non-zero turns the launch into a no-op
};

struct PackLayout {
    size_t w1, b1, w2, b2, w3, b3, total;  // offsets in floats
};

__host__ __device__ inline PackLayout pack_layout(int NT1, int NT2, int KC1) {
    PackLayout L;
    L.w1 = 0;
    L.b1 = L.w1 + (size_t)KC1 * NT1 * 32 * 32;
    L.w2 = L.b1 + (size_t)NT1 * 32;
    L.b2 = L.w2 + (size_t)NT1 * NT2 * 32 * 32;
    L.w3 = L.b2 + (size_t)NT2 * 32;
    L.b3 = L.w3 + (size_t)NT2 * 32 * 32;
    L.total = L.b3 + 32;
    return L;
}

}  // namespace smxf
using smxf::FusedArgs;
using smxf::PackLayout;
using smxf::pack_layout;

// smx_mlp3_rows16.hip: SMX_E_UNSUPPORTED when the shape / alignment is outside its fast path (the
// caller then launches the 32-row kernel)
__attribute__((visibility("hidden"))) int smx_rows16_launch(const FusedArgs& A, int H1, int H2, hipStream_t st);
