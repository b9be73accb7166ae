// Small-batch dense layers on FP32 MFMA for the PPO/DDPG epoch loops
// (surreal/learner/ppo.py:227-353: forward_actor/forward_critic on B rows, loss.backward()).
//
// These GEMMs are tiny (B = 1024 rows, K <= 1024) and latency-bound, so the decomposition is
// chosen for span, not reuse: one workgroup per 32x32 output tile, its 4 wavefronts split K
// (each wave owns every 4th 32-wide K block), operands go HBM/L2 -> registers directly in MFMA
// fragment order (everything here is L2-resident), partial tiles are combined through LDS in a
// fixed order (deterministic), and the epilogue (bias / ReLU / tanh / ReLU-mask / bias-gradient /
// sum-of-squares for clip_grad_norm_) is fused.  Up to 3 problems are grouped into one launch.
#include "smx_common.h"

namespace {

struct GemmProb {
    const float* A;
    const float* B;
    const float* bias;   // [N] or null
    const float* mask;   // [M, ldc] or null : C *= (mask > 0)
    float* C;
    float* dbias;        // [M] or null : dbias[m] = sum_k A(m,k)   (bias gradient of a dW GEMM)
    float* sumsq;        // per-tile sum of squares of C (+dbias) or null
    int lda, ldb, ldc, M, N, K;
    int a_kc, b_kc, act;
    int tiles_m, tiles_n, tile_base, a_vec, b_vec;
};

struct GemmBatch {
    GemmProb p[3];
    int n;
    const int* stop;
};

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// operand fragment for one 8-wide k group: lane (i, kh) holds X(row0+i, k0 + 4kh + r), r = 0..3
__device__ __forceinline__ float4 load_frag(const float* __restrict__ X, int ld, int kc, int vec,
                                            int row, int nrows, int k0, int K) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row >= nrows) return v;
    if (kc) {
        const float* p = X + (size_t)row * ld + k0;
        if (vec && k0 + 3 < K) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            if (k0 + 0 < K) v.x = p[0];
            if (k0 + 1 < K) v.y = p[1];
            if (k0 + 2 < K) v.z = p[2];
            if (k0 + 3 < K) v.w = p[3];
        }
    } else {
        const float* p = X + (size_t)k0 * ld + row;
        if (k0 + 0 < K) v.x = p[0];
        if (k0 + 1 < K) v.y = p[(size_t)ld];
        if (k0 + 2 < K) v.z = p[(size_t)2 * ld];
        if (k0 + 3 < K) v.w = p[(size_t)3 * ld];
    }
    return v;
}

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == SMX_ACT_RELU) return (v < 0.f) ? 0.f : v;
    if (act == SMX_ACT_TANH) return tanhf(v);
    return v;
}

__global__ __launch_bounds__(256) void gemm32_kernel(GemmBatch G) {
    if (G.stop && *G.stop) return;
    __shared__ float red[4][32 * 33];
    __shared__ float dbr[8][32];
    __shared__ float sred[16];

    int pi = 0;
    if (G.n > 1 && (int)blockIdx.x >= G.p[1].tile_base) pi = 1;
    if (G.n > 2 && (int)blockIdx.x >= G.p[2].tile_base) pi = 2;
    const GemmProb& P = G.p[pi];
    const int tile = blockIdx.x - P.tile_base;
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.f;
    float asum = 0.f;

    const int nsb = (P.K + 31) >> 5;  // 32-wide K super-blocks; wave wv owns sb = wv, wv+4, ...
    float4 a_cur[4], b_cur[4], a_nxt[4], b_nxt[4];
    {
        const int kb = wv * 32 + 4 * kh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a_cur[q] = load_frag(P.A, P.lda, P.a_kc, P.a_vec, m0 + i, P.M, kb + 8 * q, P.K);
            b_cur[q] = load_frag(P.B, P.ldb, P.b_kc, P.b_vec, n0 + i, P.N, kb + 8 * q, P.K);
        }
    }
    for (int sb = wv; sb < nsb; sb += 4) {
        const int kn = (sb + 4) * 32 + 4 * kh;  // past K -> guarded loads return zeros
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a_nxt[q] = load_frag(P.A, P.lda, P.a_kc, P.a_vec, m0 + i, P.M, kn + 8 * q, P.K);
            b_nxt[q] = load_frag(P.B, P.ldb, P.b_kc, P.b_vec, n0 + i, P.N, kn + 8 * q, P.K);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = MFMA32(a_cur[q].x, b_cur[q].x, acc);
            acc = MFMA32(a_cur[q].y, b_cur[q].y, acc);
            acc = MFMA32(a_cur[q].z, b_cur[q].z, acc);
            acc = MFMA32(a_cur[q].w, b_cur[q].w, acc);
            asum += (a_cur[q].x + a_cur[q].y) + (a_cur[q].z + a_cur[q].w);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { a_cur[q] = a_nxt[q]; b_cur[q] = b_nxt[q]; }
    }

    // ---- combine the 4 K-partials through LDS (fixed order => deterministic) -------------
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int r = (s & 3) + 8 * (s >> 2) + 4 * kh;
        red[wv][r * 33 + i] = acc[s];
    }
    dbr[wv * 2 + kh][i] = asum;
    __syncthreads();

    const int orow = tid >> 3, oc0 = (tid & 7) * 4;
    const int m = m0 + orow;
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = n0 + oc0 + e;
        float v = ((red[0][orow * 33 + oc0 + e] + red[1][orow * 33 + oc0 + e]) +
                   red[2][orow * 33 + oc0 + e]) + red[3][orow * 33 + oc0 + e];
        if (m < P.M && n < P.N) {
            if (P.bias) v += P.bias[n];
            v = act_f(v, P.act);
            if (P.mask) v = (P.mask[(size_t)m * P.ldc + n] > 0.f) ? v : 0.f;
            P.C[(size_t)m * P.ldc + n] = v;
            ss += v * v;
        }
    }
    if (P.dbias && tn == 0 && tid < 32) {
        float d = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) d += dbr[w][tid];
        if (m0 + tid < P.M) {
            P.dbias[m0 + tid] = d;
            ss += d * d;
        }
    }
    if (P.sumsq) {
        const float t = smx_block_sum(ss, sred);
        if (tid == 0) P.sumsq[tile] = t;
    }
}

inline void fill_prob(GemmProb& P, const float* A, int lda, int a_kc, const float* B, int ldb,
                      int b_kc, const float* bias, const float* mask, float* C, int ldc, int M,
                      int N, int K, int act, float* dbias, float* sumsq, int tile_base) {
    P.A = A; P.B = B; P.bias = bias; P.mask = mask; P.C = C; P.dbias = dbias; P.sumsq = sumsq;
    P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.M = M; P.N = N; P.K = K;
    P.a_kc = a_kc; P.b_kc = b_kc; P.act = act;
    P.tiles_m = (M + 31) / 32; P.tiles_n = (N + 31) / 32; P.tile_base = tile_base;
    P.a_vec = a_kc && (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    P.b_vec = b_kc && (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
}

inline int launch_batch(GemmBatch& G, hipStream_t st) {
    const GemmProb& L = G.p[G.n - 1];
    const int blocks = L.tile_base + L.tiles_m * L.tiles_n;
    hipLaunchKernelGGL(gemm32_kernel, dim3(blocks), dim3(256), 0, st, G);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMX_OK : (int)e;
}

}  // namespace

extern "C" int smx_linear_f32(const float* A, int32_t lda, int32_t a_kcontig, const float* B,
                              int32_t ldb, int32_t b_kcontig, const float* bias, float* C,
                              int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t act,
                              const float* relu_mask, const int32_t* stop_flag,
                              smx_stream_t stream) {
    SMX_REQUIRE(A && B && C, SMX_E_NULL);
    SMX_REQUIRE(M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N, SMX_E_SHAPE);
    GemmBatch G;
    G.n = 1;
    G.stop = stop_flag;
    fill_prob(G.p[0], A, lda, a_kcontig, B, ldb, b_kcontig, bias, relu_mask, C, ldc, M, N, K, act,
              nullptr, nullptr, 0);
    return launch_batch(G, smx_s(stream));
}

// dW[M,N] = dZ^T . X (dZ [rows, >=M] stride ldz, X [rows, >=N] stride ldx), db[M] = column sums of dZ
extern "C" int smx_linear_wgrad_f32(const float* dZ, int32_t ldz, const float* X, int32_t ldx,
                                    float* dW, int32_t ldw, float* db, int32_t M, int32_t N,
                                    int32_t rows, smx_stream_t stream) {
    SMX_REQUIRE(dZ && X && dW, SMX_E_NULL);
    SMX_REQUIRE(M > 0 && N > 0 && rows > 0 && ldz >= M && ldx >= N && ldw >= N, SMX_E_SHAPE);
    GemmBatch G;
    G.n = 1;
    G.stop = nullptr;
    fill_prob(G.p[0], dZ, ldz, 0, X, ldx, 0, nullptr, nullptr, dW, ldw, M, N, rows, SMX_ACT_NONE, db,
              nullptr, 0);
    return launch_batch(G, smx_s(stream));
}

extern "C" int smx_mlp3_forward_f32(const smx_mlp3_t* net, const float* x, int64_t rows, float* h1,
                                    float* h2, float* out, int32_t out_act,
                                    const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(net && x && h1 && h2 && out, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && rows < (1 << 30), SMX_E_SHAPE);
    const int R = (int)rows;
    int rc;
    rc = smx_linear_f32(x, net->D, 1, net->W1, net->D, 1, net->b1, h1, net->H1, R, net->H1, net->D,
                        SMX_ACT_RELU, nullptr, stop_flag, stream);
    if (rc) return rc;
    rc = smx_linear_f32(h1, net->H1, 1, net->W2, net->H1, 1, net->b2, h2, net->H2, R, net->H2,
                        net->H1, SMX_ACT_RELU, nullptr, stop_flag, stream);
    if (rc) return rc;
    return smx_linear_f32(h2, net->H2, 1, net->W3, net->H2, 1, net->b3, out, net->OUT, R, net->OUT,
                          net->H2, out_act, nullptr, stop_flag, stream);
}

extern "C" int32_t smx_mlp3_backward_partials(int32_t D, int32_t H1, int32_t H2, int32_t OUT) {
    auto t = [](int a) { return (a + 31) / 32; };
    return t(H1) * t(D) + t(H2) * t(H1) + t(OUT) * t(H2);
}

extern "C" int smx_mlp3_backward_f32(const smx_mlp3_t* net, const float* x, const float* h1,
                                     const float* h2, const float* dz3, int64_t rows, float* dz2,
                                     float* dz1, float* grads, float* sumsq_partials,
                                     const int32_t* stop_flag, smx_stream_t stream) {
    SMX_REQUIRE(net && x && h1 && h2 && dz3 && dz2 && dz1 && grads, SMX_E_NULL);
    SMX_REQUIRE(rows > 0 && rows < (1 << 30), SMX_E_SHAPE);
    const int R = (int)rows, D = net->D, H1 = net->H1, H2 = net->H2, O = net->OUT;
    int rc;
    // dz2 = (dz3 . W3) * relu'(h2)      [R, H2], K = OUT
    rc = smx_linear_f32(dz3, O, 1, net->W3, H2, 0, nullptr, dz2, H2, R, H2, O, SMX_ACT_NONE, h2,
                        stop_flag, stream);
    if (rc) return rc;
    // dz1 = (dz2 . W2) * relu'(h1)      [R, H1], K = H2
    rc = smx_linear_f32(dz2, H2, 1, net->W2, H1, 0, nullptr, dz1, H1, R, H1, H2, SMX_ACT_NONE, h1,
                        stop_flag, stream);
    if (rc) return rc;
    // dW_l = dz_l^T . input_l , db_l = column sums of dz_l : three problems, one launch
    float* gW1 = grads;
    float* gb1 = gW1 + (size_t)H1 * D;
    float* gW2 = gb1 + H1;
    float* gb2 = gW2 + (size_t)H2 * H1;
    float* gW3 = gb2 + H2;
    float* gb3 = gW3 + (size_t)O * H2;
    GemmBatch G;
    G.n = 3;
    G.stop = stop_flag;
    fill_prob(G.p[0], dz1, H1, 0, x, D, 0, nullptr, nullptr, gW1, D, H1, D, R, SMX_ACT_NONE, gb1,
              sumsq_partials, 0);
    int base = G.p[0].tiles_m * G.p[0].tiles_n;
    fill_prob(G.p[1], dz2, H2, 0, h1, H1, 0, nullptr, nullptr, gW2, H1, H2, H1, R, SMX_ACT_NONE, gb2,
              sumsq_partials ? sumsq_partials + base : nullptr, base);
    base += G.p[1].tiles_m * G.p[1].tiles_n;
    fill_prob(G.p[2], dz3, O, 0, h2, H2, 0, nullptr, nullptr, gW3, H2, O, H2, R, SMX_ACT_NONE, gb3,
              sumsq_partials ? sumsq_partials + base : nullptr, base);
    return launch_batch(G, smx_s(stream));
}
