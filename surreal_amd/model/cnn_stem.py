"""
CNN stem of the PPO model (surreal/model/model_builders/builders.py:8-33 ``CNNStemNetwork``):
Conv2d(16, k8, s4)-ReLU-Conv2d(32, k4, s2)-ReLU-Flatten-Linear(cnn_feature_dim)-ReLU applied to the
camera frames scaled by 1/255 (ppo_net.py:268-273, 368-375).

Both convolutions are GEMMs over patch rows on the FP32-MFMA layer kernel; activations are kept
channel-last ``[frame, pixel, channel]`` (what those GEMMs write).  torch flattens channel-first, so
the Linear runs against a re-indexed copy of its weight (``smx_flatten_order_f32``).
"""
import collections
import types

import numpy as np
import torch

from surreal_amd import _lib as L


class CnnParams(object):
    """views of the stem's parameters inside a flat buffer, torch layouts:
    conv1.W [c1, C, k1, k1], conv1.b, conv2.W [c2, c1, k2, k2], conv2.b, fc.W [feat, c2*P2], fc.b"""

    def __init__(self, flat, offset, in_shape, feat_dim, conv_channels=(16, 32),
                 kernel_sizes=(8, 4), strides=(4, 2)):
        C, H, W = [int(v) for v in in_shape]
        self.C, self.H, self.W, self.feat = C, H, W, int(feat_dim)
        (self.c1, self.c2), (self.k1, self.k2), (self.s1, self.s2) = conv_channels, kernel_sizes, strides
        self.H1, self.W1 = (H - self.k1) // self.s1 + 1, (W - self.k1) // self.s1 + 1
        self.H2, self.W2 = (self.H1 - self.k2) // self.s2 + 1, (self.W1 - self.k2) // self.s2 + 1
        if self.H1 <= 0 or self.W1 <= 0 or self.H2 <= 0 or self.W2 <= 0:
            raise ValueError('camera frames %r are too small for the CNN stem' % (in_shape,))
        self.P1, self.P2 = self.H1 * self.W1, self.H2 * self.W2
        self.K1, self.K2 = C * self.k1 * self.k1, self.c1 * self.k2 * self.k2
        self.flat_dim = self.c2 * self.P2
        sizes = [('conv1.W', (self.c1, C, self.k1, self.k1)), ('conv1.b', (self.c1,)),
                 ('conv2.W', (self.c2, self.c1, self.k2, self.k2)), ('conv2.b', (self.c2,)),
                 ('fc.W', (self.feat, self.flat_dim)), ('fc.b', (self.feat,))]
        self.views = collections.OrderedDict()
        o = offset
        for name, shp in sizes:
            n = int(np.prod(shp))
            self.views[name] = flat[o:o + n].view(*shp)
            o += n
        self.numel = o - offset
        self.offset = offset

    @staticmethod
    def count(in_shape, feat_dim, conv_channels=(16, 32), kernel_sizes=(8, 4), strides=(4, 2)):
        C, H, W = [int(v) for v in in_shape]
        h1, w1 = (H - kernel_sizes[0]) // strides[0] + 1, (W - kernel_sizes[0]) // strides[0] + 1
        h2, w2 = (h1 - kernel_sizes[1]) // strides[1] + 1, (w1 - kernel_sizes[1]) // strides[1] + 1
        c1, c2 = conv_channels
        n = c1 * C * kernel_sizes[0] ** 2 + c1 + c2 * c1 * kernel_sizes[1] ** 2 + c2
        n += feat_dim * c2 * h2 * w2 + feat_dim
        return (n + 3) & ~3          # keeps whatever follows 16-byte aligned

    def init_torch_default(self):
        """torch.nn.Conv2d / Linear default init (kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)));
        torchx's own init is unknown (source absent) -- parity tests inject parameters"""
        for name, v in self.views.items():
            w = self.views[name.split('.')[0] + '.W']
            fan_in = int(np.prod(w.shape[1:]))
            v.uniform_(-1.0 / np.sqrt(fan_in), 1.0 / np.sqrt(fan_in))


class CnnStem(object):
    """forward / backward of the stem over F frames through the kernel facade"""

    def __init__(self, kernels):
        self.K = kernels

    @staticmethod
    def workspace(p, F, device, backward=True):
        f = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)  # noqa: E731
        ws = types.SimpleNamespace(F=F)
        # the first convolution's patch matrix (2.2 GB at 7168 camera frames) is only allocated if something asks for
        # it: uint8 frames inside the implicit-GEMM kernels' limits never do (_cols1)
        ws.cols1, ws.y1 = None, f(F * p.P1, p.c1)
        ws.cols1_shape, ws.device = (F * p.P1, p.K1), device
        ws.cols2, ws.y2 = None, f(F * p.P2, p.c2)
        ws.cols2_shape = (F * p.P2, p.K2)
        ws.wfc = f(p.feat, p.flat_dim)                 # fc.W re-indexed channel-last
        ws.sk = None
        if backward:
            ws.dy2, ws.dcols2, ws.dy1 = f(F * p.P2, p.c2), None, f(F * p.P1, p.c1)
            ws.gwfc = f(p.feat, p.flat_dim)
        return ws

    @staticmethod
    def _cols1(ws):
        if ws.cols1 is None:
            ws.cols1 = torch.empty(*ws.cols1_shape, device=ws.device, dtype=torch.float32)
        return ws.cols1

    @staticmethod
    def _cols2(ws):
        if ws.cols2 is None:
            ws.cols2 = torch.empty(*ws.cols2_shape, device=ws.device, dtype=torch.float32)
        return ws.cols2

    def splitk_workspace(self, p, F, device):
        """split-K workspace of the two convolution weight gradients (their GEMMs sum over
        F * pixels rows)"""
        n = max(self.K.linear_wgrad_ws_floats(p.c2, p.K2, F * p.P2),
                self.K.linear_wgrad_ws_floats(p.c1, p.K1, F * p.P1),
                self.K.conv_u8_wgrad_ws_floats(p.c1, p.K1), self.K.conv_cl_wgrad_ws_floats(p.c2, p.k2))
        return torch.empty(n, device=device, dtype=torch.float32) if n else None

    def forward(self, p, frames, F, ws, out, stop=None, cols1_tag=None):
        """frames: uint8 or fp32 [F, C, H, W] (contiguous); out: [F, feat] view (any row stride).
        cols1_tag: identifies the CONTENT of `frames` (the caller's choice, e.g. (learn counter, pointer, F)); when it
        equals the tag of the patch matrix ws.cols1 already holds, the first convolution's im2col is skipped -- the
        patches do not depend on the weights, and a learn runs ~22 forwards over the same frames."""
        K, v = self.K, p.views
        if K.conv_u8_supported(frames, p.C, p.H, p.W, p.k1, p.s1, p.c1, v['conv1.W']):
            # implicit GEMM straight from the uint8 frames: no patch matrix on the forward path.  The weight
            # gradient of this layer still reads one (backward() builds it on first use, once per set of frames)
            K.conv_u8_forward(frames, F, p.C, p.H, p.W, p.k1, p.s1, v['conv1.W'], v['conv1.b'], p.c1, ws.y1,
                              stop=stop)
            ws.cols1_src = (frames, F, cols1_tag)
        else:
            if cols1_tag is None or getattr(ws, 'cols1_tag', None) != cols1_tag:
                K.im2col(frames, F, p.C, p.H, p.W, p.k1, p.s1, self._cols1(ws), scale_div=255.0)
                ws.cols1_tag = cols1_tag
            ws.cols1_src = None
            K.linear(ws.cols1, 1, v['conv1.W'].view(p.c1, p.K1), 1, v['conv1.b'], ws.y1, F * p.P1, p.c1,
                     p.K1, act=L.SMX_ACT_RELU, stop=stop)
        ws.conv2_implicit = K.conv_cl_supported(ws.y1, p.c1, p.k2, p.c2)
        if ws.conv2_implicit:          # gathers from the channel-last y1: no patch matrix for this layer either
            K.conv_cl_forward(ws.y1, F, p.c1, p.H1, p.W1, p.k2, p.s2, v['conv2.W'], v['conv2.b'], p.c2, ws.y2, stop=stop)
        else:
            K.im2col(ws.y1, F, p.c1, p.H1, p.W1, p.k2, p.s2, self._cols2(ws), channel_last=True)
            K.linear(ws.cols2, 1, v['conv2.W'].view(p.c2, p.K2), 1, v['conv2.b'], ws.y2, F * p.P2, p.c2,
                     p.K2, act=L.SMX_ACT_RELU, stop=stop)
        K.flatten_order(v['fc.W'], p.feat, p.c2, p.P2, True, ws.wfc)
        K.linear(ws.y2[:F * p.P2].view(F, p.flat_dim), 1, ws.wfc, 1, v['fc.b'], out, F, p.feat, p.flat_dim,
                 act=L.SMX_ACT_RELU, ldc=out.stride(0), stop=stop)

    def backward(self, p, F, ws, dfeat, grads, stop=None):
        """dfeat: [F, feat] view = dLoss/d(stem output), ALREADY multiplied by the output ReLU's
        mask; ws must still hold this pass's forward activations.  grads: flat, parameter layout."""
        K, v = self.K, p.views
        g = CnnParams.__new__(CnnParams)             # same layout over the gradient buffer
        g.views = collections.OrderedDict()
        o = 0
        for name, t in v.items():
            g.views[name] = grads[o:o + t.numel()].view(t.shape)
            o += t.numel()
        gv = g.views
        ldz = dfeat.stride(0)
        # Linear: dW (channel-last, then back to torch's order), db, d(flat) * relu'(y2)
        y2 = ws.y2[:F * p.P2].view(F, p.flat_dim)         # F may be a tail chunk of the workspace
        dy2 = ws.dy2[:F * p.P2].view(F, p.flat_dim)
        K.linear_wgrad(dfeat, y2, ws.gwfc, gv['fc.b'], p.feat, p.flat_dim, F, ldz=ldz)
        K.flatten_order(ws.gwfc, p.feat, p.c2, p.P2, False, gv['fc.W'])
        K.linear(dfeat, 1, ws.wfc, 0, None, dy2, F, p.flat_dim, p.feat, relu_mask=y2, lda=ldz, stop=stop)
        # conv2: dW, db, data gradient scattered back through the patches * relu'(y1)
        if getattr(ws, 'conv2_implicit', False) and ws.sk is not None:
            K.conv_cl_wgrad(ws.y1, F, p.c1, p.H1, p.W1, p.k2, p.s2, ws.dy2, p.c2, gv['conv2.W'].view(p.c2, p.K2),
                            gv['conv2.b'], ws.sk, stop=stop)
        else:
            if getattr(ws, 'conv2_implicit', False):        # (no split-K workspace: the materialised route)
                K.im2col(ws.y1, F, p.c1, p.H1, p.W1, p.k2, p.s2, self._cols2(ws), channel_last=True)
            K.linear_wgrad(ws.dy2, ws.cols2, gv['conv2.W'].view(p.c2, p.K2), gv['conv2.b'], p.c2, p.K2,
                           F * p.P2, ws=ws.sk)
        if K.conv_cl_dgrad_supported(ws.dy2, p.c1, p.k2, p.s2, p.c2):
            # d(y1) * relu'(y1) straight from dy2 and the weight: no dcols matrix, no col2im pass
            K.conv_cl_dgrad(ws.dy2, F, p.c1, p.H1, p.W1, p.k2, p.s2, v['conv2.W'], p.c2, ws.y1, ws.dy1, stop=stop)
        else:
            if ws.dcols2 is None:
                ws.dcols2 = torch.empty(*ws.cols2_shape, device=ws.device, dtype=torch.float32)
            K.linear(ws.dy2, 1, v['conv2.W'].view(p.c2, p.K2), 0, None, ws.dcols2, F * p.P2, p.K2, p.c2,
                     stop=stop)
            K.col2im(ws.dcols2, F, p.c1, p.H1, p.W1, p.k2, p.s2, ws.y1, ws.dy1)
        # conv1: dW, db (the frames carry no gradient).  Frames that went through the implicit-GEMM forward also give
        # their weight gradient without a patch matrix; otherwise it is the GEMM over the materialised patches
        src = getattr(ws, 'cols1_src', None)
        if src is not None and ws.sk is not None:
            frames, Fs, _ = src
            K.conv_u8_wgrad(frames, Fs, p.C, p.H, p.W, p.k1, p.s1, ws.dy1, p.c1, gv['conv1.W'].view(p.c1, p.K1),
                            gv['conv1.b'], ws.sk, stop=stop)
            return
        if src is not None:
            frames, Fs, tag = src
            if tag is None or getattr(ws, 'cols1_tag', None) != tag:
                K.im2col(frames, Fs, p.C, p.H, p.W, p.k1, p.s1, self._cols1(ws), scale_div=255.0)
                ws.cols1_tag = tag
        K.linear_wgrad(ws.dy1, ws.cols1, gv['conv1.W'].view(p.c1, p.K1), gv['conv1.b'], p.c1, p.K1,
                       F * p.P1, ws=ws.sk)
