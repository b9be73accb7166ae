// smx_wgrad.hip: split-K weight gradients over many rows straight from the row-major operands (no LDS staging).
#pragma once
#include "smx_common.h"

constexpr int WGRAD_MAX_PROBS = 4;
constexpr long SMX_WGRAD_ROWS_MIN = 32768;     // below: too few row chunks of a useful length for one workgroup per CU

struct WgradProb {
    const float* A;      // dZ [rows, >= M], row stride lda
    const float* B;      // X  [rows, >= N], row stride ldb
    float* Cpart;        // [splits][M][ldc] partial matrices: this problem writes columns [0, N) of each row (ldc >= N;
                         // a column group of a wider dW: Cpart points at the group's first column)
    int ldc;             // row stride of a partial matrix; the partial matrices are c_split floats apart
    long c_split;
    float* bpart;        // [splits][M] partial column sums of dZ, or null
    int M, N, lda, ldb, rows;
    int b_col0;          // columns of X in front of B (already folded into the pointer: bounds the buffer range)
    int splits, k_chunk; // rows [s k_chunk, (s + 1) k_chunk) belong to split s; k_chunk % 32 == 0
    int wm, wn;          // the workgroup's 8 wavefronts as wm x wn blocks of tiles (filled by smx_wgrad_rows_launch)
    int wg_base;
};
struct WgradBatch {
    WgradProb p[WGRAD_MAX_PROBS];
    int n;
    const int* stop;
};

__attribute__((visibility("hidden"))) bool smx_wgrad_rows_eligible(const float* A, int lda, const float* B, int ldb, int M, int N,
                                                                 long rows);
__attribute__((visibility("hidden"))) int smx_wgrad_rows_launch(WgradBatch& G, hipStream_t st);
// shape-only part of the eligibility test (what sizes the split-K workspace)
__attribute__((visibility("hidden"))) bool smx_wgrad_rows_plan(int M, int N, int* wm, int* wn);
// a dW too wide for one workgroup's registers: N cut into `groups` column groups of `width` columns (the last one narrower),
// each a problem of its own that re-reads dZ; 0 when no cut fits
__attribute__((visibility("hidden"))) int smx_wgrad_rows_groups(int M, int N, int* width);
