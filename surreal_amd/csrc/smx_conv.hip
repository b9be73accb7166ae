// CNN stem pieces (surreal/model/model_builders/builders.py:8-33: Conv2d(16, k8, s4)-ReLU-
// Conv2d(32, k4, s2)-ReLU-Flatten-Linear(cnn_feature_dim)-ReLU on the camera image scaled by
// 1/255, ppo_net.py:368-375).  The convolutions run as GEMMs on the FP32-MFMA layer kernel
// (smx_linear_f32 / smx_linear_wgrad_f32); this file holds the data movement around them:
//   * im2col: patches -> rows, reading the uint8 camera frames directly (the reference first
//     converts the whole batch to fp32, ppo.py:436-441) and applying x / 255 on the way;
//   * col2im: the data gradient of a convolution as a GATHER over the (kernel/stride)^2 output
//     positions that saw each input element -- deterministic, no atomics -- with the ReLU mask of
//     the producing layer fused;
//   * the Flatten order: activations are kept channel-last [frame, pixel, channel] (that is what
//     a GEMM over patch rows writes), torch flattens channel-first, so the Linear's weight is
//     re-indexed [out, c*P + p] <-> [out, p*C + c].
#include "smx_common.h"

namespace {

struct ConvGeom {
    int C, Hin, Win, kh, kw, stride, Ho, Wo;
};

// SRC_U8: src is uint8 NCHW frames, value = float(u8) / 255.0f; else fp32.
// CHANNEL_LAST: src is [F, Hin*Win, C] (an activation of ours); else [F, C, Hin, Win].
template <bool SRC_U8, bool CHANNEL_LAST>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ src, ConvGeom g,
                                                     long long total, float scale_div,
                                                     float* __restrict__ cols) {
    const int K = g.C * g.kh * g.kw;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const long long row = idx / K;
        const int k = (int)(idx - row * K);
        const int c = k / (g.kh * g.kw), ij = k - c * (g.kh * g.kw);
        const int i = ij / g.kw, j = ij - i * g.kw;
        const long long f = row / (g.Ho * g.Wo);
        const int p = (int)(row - f * (g.Ho * g.Wo));
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        const int y = oy * g.stride + i, x = ox * g.stride + j;
        const long long s = CHANNEL_LAST
                                ? ((f * g.Hin + y) * g.Win + x) * g.C + c
                                : ((f * g.C + c) * g.Hin + y) * (long long)g.Win + x;
        float v = SRC_U8 ? (float)static_cast<const unsigned char*>(src)[s]
                         : static_cast<const float*>(src)[s];
        if (scale_div != 0.f) v = v / scale_div;
        cols[idx] = v;
    }
}

// dX[f, y, x, c] = mask * sum over (i, j) with oy*stride + i == y, ox*stride + j == x of
//                  dcols[(f, oy, ox), c*kh*kw + i*kw + j]          (channel-last dX)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, ConvGeom g,
                                                     long long total,
                                                     const float* __restrict__ relu_of,
                                                     float* __restrict__ dx) {
    const int K = g.C * g.kh * g.kw;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % g.C);
        const long long pix = idx / g.C;
        const long long f = pix / (g.Hin * g.Win);
        const int yx = (int)(pix - f * (g.Hin * g.Win));
        const int y = yx / g.Win, x = yx - y * g.Win;
        float acc = 0.f;
        for (int i = y % g.stride; i < g.kh; i += g.stride) {
            const int oy = (y - i) / g.stride;
            if (y - i < 0 || oy >= g.Ho) continue;
            for (int j = x % g.stride; j < g.kw; j += g.stride) {
                const int ox = (x - j) / g.stride;
                if (x - j < 0 || ox >= g.Wo) continue;
                acc += dcols[((f * g.Ho + oy) * g.Wo + ox) * K + c * (g.kh * g.kw) + i * g.kw + j];
            }
        }
        if (relu_of) acc = (relu_of[idx] > 0.f) ? acc : 0.f;
        dx[idx] = acc;
    }
}

// to_channel_last != 0: out[o, p*C + c] = in[o, c*P + p] ; else the inverse
__global__ __launch_bounds__(256) void flatten_order_kernel(const float* __restrict__ in, int O,
                                                            int C, int P, int to_channel_last,
                                                            float* __restrict__ out) {
    const long long total = (long long)O * C * P;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const long long o = idx / (C * P);
        const int r = (int)(idx - o * (C * P));
        if (to_channel_last) {
            const int p = r / C, c = r - p * C;
            out[idx] = in[o * (C * P) + c * P + p];
        } else {
            const int c = r / P, p = r - c * P;
            out[idx] = in[o * (C * P) + p * C + c];
        }
    }
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

inline bool geom_ok(const ConvGeom& g) {
    return g.C > 0 && g.Hin > 0 && g.Win > 0 && g.kh > 0 && g.kw > 0 && g.stride > 0 &&
           g.Ho == (g.Hin - g.kh) / g.stride + 1 && g.Wo == (g.Win - g.kw) / g.stride + 1 &&
           g.Ho > 0 && g.Wo > 0;
}

}  // namespace

extern "C" int smx_im2col_f32(const void* src, int32_t src_is_u8, int32_t channel_last, int64_t F,
                              int32_t C, int32_t Hin, int32_t Win, int32_t kh, int32_t kw,
                              int32_t stride, float scale_div, float* cols, smx_stream_t stream) {
    SMX_REQUIRE(src && cols, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, kh, kw, stride, (Hin - kh) / stride + 1, (Win - kw) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g), SMX_E_SHAPE);
    SMX_REQUIRE(!(src_is_u8 && channel_last), SMX_E_UNSUPPORTED);
    const long long total = (long long)F * g.Ho * g.Wo * C * kh * kw;
    const int blocks = grid_for(total);
    if (src_is_u8)
        hipLaunchKernelGGL((im2col_kernel<true, false>), dim3(blocks), dim3(256), 0, smx_s(stream),
                           src, g, total, scale_div, cols);
    else if (channel_last)
        hipLaunchKernelGGL((im2col_kernel<false, true>), dim3(blocks), dim3(256), 0, smx_s(stream),
                           src, g, total, scale_div, cols);
    else
        hipLaunchKernelGGL((im2col_kernel<false, false>), dim3(blocks), dim3(256), 0,
                           smx_s(stream), src, g, total, scale_div, cols);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_col2im_f32(const float* dcols, int64_t F, int32_t C, int32_t Hin, int32_t Win,
                              int32_t kh, int32_t kw, int32_t stride, const float* relu_of,
                              float* dx, smx_stream_t stream) {
    SMX_REQUIRE(dcols && dx, SMX_E_NULL);
    ConvGeom g{C, Hin, Win, kh, kw, stride, (Hin - kh) / stride + 1, (Win - kw) / stride + 1};
    SMX_REQUIRE(F > 0 && geom_ok(g), SMX_E_SHAPE);
    const long long total = (long long)F * Hin * Win * C;
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total)), dim3(256), 0, smx_s(stream), dcols, g,
                       total, relu_of, dx);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_flatten_order_f32(const float* in, int32_t O, int32_t C, int32_t P,
                                     int32_t to_channel_last, float* out, smx_stream_t stream) {
    SMX_REQUIRE(in && out, SMX_E_NULL);
    SMX_REQUIRE(O > 0 && C > 0 && P > 0, SMX_E_SHAPE);
    hipLaunchKernelGGL(flatten_order_kernel, dim3(grid_for((long long)O * C * P)), dim3(256), 0,
                       smx_s(stream), in, O, C, P, to_channel_last, out);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}
