// smx_mlp3_bwd16.hip: dz3 -> dz2 -> dz1 -> dx of the 3-layer MLP over many rows in one launch
#pragma once
#include "smx_common.h"

struct Bwd16Args {
    const float* pt1;   // packed transposed weights (filled by the launcher)
    const float* pt2;
    const float* pt3;
    const float* dz3;   // [rows, OUT], row stride ld3
    int ld3;
    const float* h1;    // saved activations (the ReLU masks)
    const float* h2;
    float* dz2;         // [rows, H2]
    float* dz1;         // [rows, H1]
    float* dx;          // [rows, D] or null (no gradient wanted for the input)
    long rows;
    int D, H1, H2, OUT;
    const int* stop;
};

// SMX_E_UNSUPPORTED outside the kernel's shapes (the caller then runs the layered GEMMs)
__attribute__((visibility("hidden"))) int smx_mlp3_dgrad_rows_launch(const smx_mlp3_t* net, const float* h1, const float* h2,
                                                                    const float* dz3, int64_t rows, float* dz2, float* dz1,
                                                                    float* dx, float* packedT, int64_t packedT_floats,
                                                                    const int32_t* stop_flag, hipStream_t st);
