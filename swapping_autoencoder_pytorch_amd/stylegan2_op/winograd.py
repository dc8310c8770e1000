"""Winograd F(2x2, 3x3) route for the wide 3x3 stride-1 convolutions (pad 1 or 0) (csrc/winograd.hip, include/sae_hip.h: sae_wino_*).

    y = output_transform( 16 x [ conv1x1( input_transform(x)[xi], U[xi] ) ] )        U = weight_transform(w)

The same three C-ABI transforms serve the forward (``F.conv2d`` at models/networks/stylegan2_layers.py:136,315) and the data
gradient (the reversed filter with the channel roles swapped); the 16 transform-domain products are one batched launch of the 1x1
MFMA gather.  ``wgrad`` is the weight gradient on the same sixteen points.  2.25x fewer multiplications for two extra passes over 4x the activation: it pays on layers with many channels on
small maps.  OFF unless ``SAE_WINOGRAD=1`` -- built and verified against the oracle this round (CPU emulator), not yet measured
on the GPU; ``SAE_WINOGRAD_MIN_C`` (default 256) is the smallest channel count that takes it.
Results differ from the direct kernels' by rounding (~1e-6 relative), so the route is never mixed into bit-identity checks."""
import os

import torch

from .. import hip_lib


def enabled():
    return os.environ.get("SAE_WINOGRAD", "0") == "1"


def min_channels():
    return int(os.environ.get("SAE_WINOGRAD_MIN_C", "256"))


def eligible(geom):
    """3x3, stride 1, pad 1 or 0, even sides (whole 2x2 output tiles), wide enough, exact-fp32 arithmetic."""
    if not enabled() or geom.k != 3 or geom.stride != 1 or geom.pad not in (0, 1) or (geom.h & 1) or (geom.w & 1):
        return False
    if min(geom.c, geom.m) < min_channels():
        return False
    return hip_lib.get().query("get_conv_math") == 0


def conv(x, w, geom, transpose=False, bias=None, act=None, x_scale=None, row_scale=None, col_scale=None, out_scale=None,
         noise=None, noise_weight=None):
    """alpha * conv(x * x_scale, w') (transpose=False: x is [n, c, h, w]) or alpha * conv^T(gy * x_scale, w') (transpose=True: x
    is the output gradient [n, m, h, w]) with w' = w * row_scale * col_scale along the OUTPUT / CONTRACTION axes of the product
    that is computed; times out_scale [n, outputs] if given; then, with act = (slope, scale),
    lrelu((. + noise_weight * noise[n]) + bias) * scale (noise, bias optional)."""
    lib = hip_lib.get()
    x = x.contiguous()
    w = w.contiguous()
    opt = [None if t is None else t.contiguous() for t in (bias, x_scale, row_scale, col_scale, out_scale, noise, noise_weight)]
    bias, x_scale, row_scale, col_scale, out_scale, noise, noise_weight = opt
    lib.check(x, w, *opt)
    d = geom.desc()
    if transpose:
        cin, cout, sm, sc, flip = geom.m, geom.c, d.w_stride_c, d.w_stride_m, 1
    else:
        cin, cout, sm, sc, flip = geom.c, geom.m, d.w_stride_m, d.w_stride_c, 0
    # forward: the layer's input with its own padding; data gradient: the output gradient with padding 2 - pad (a valid layer's
    # gradient is a full correlation)
    n = geom.n
    ih, iw, pad, h, wd = (geom.oh, geom.ow, 2 - geom.pad, geom.h, geom.w) if transpose else (geom.h, geom.w, geom.pad, geom.oh, geom.ow)
    if tuple(x.shape) != (n, cin, ih, iw):
        raise hip_lib.SaeError("winograd conv: input %s, expected (%d, %d, %d, %d)" % (tuple(x.shape), n, cin, ih, iw))
    th, tw = h // 2, wd // 2
    tiles = th * tw
    stream = lib.stream(x)
    dev = x.device
    u = torch.empty((16, cout, cin), dtype=torch.float32, device=dev)
    lib.call("wino_weights_f32", w.data_ptr(), hip_lib.ptr(row_scale), hip_lib.ptr(col_scale), u.data_ptr(), cout, cin, sm, sc,
             flip, geom.alpha, stream)
    v = torch.empty((16, n * cin, tiles), dtype=torch.float32, device=dev)
    lib.call("wino_input_f32", x.data_ptr(), hip_lib.ptr(x_scale), v.data_ptr(), n * cin, ih, iw, pad, stream)
    md = torch.empty((16, n * cout, tiles), dtype=torch.float32, device=dev)
    n_ws = lib.query("wino_gemm_workspace", n, cin, cout, th, tw)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
    lib.call("wino_gemm_f32", v.data_ptr(), u.data_ptr(), md.data_ptr(), n, cin, cout, th, tw, ws.data_ptr(), n_ws, stream)
    y = torch.empty((n, cout, h, wd), dtype=torch.float32, device=dev)
    slope, scale = act if act is not None else (0.0, 1.0)
    lib.call("wino_output_f32", md.data_ptr(), hip_lib.ptr(out_scale), hip_lib.ptr(noise), hip_lib.ptr(noise_weight),
             hip_lib.ptr(bias), y.data_ptr(), n * cout, cout, h, wd, 1 if act is not None else 0, float(slope), float(scale), stream)
    return y


def wgrad(x, gy, geom, out=None, x_scale=None, y_scale=None):
    """alpha * sum over images and pixels of (gy * y_scale) (x) (x * x_scale) as the layer's weight gradient (geom.weight_shape()),
    on the sixteen points: gw = G^T [ sum (A e A^T) o (B^T d B) ] G.  out: an existing tensor of that shape to write into (a
    slot of an armed gradient bucket)."""
    lib = hip_lib.get()
    x = x.contiguous()
    gy = gy.contiguous()
    x_scale = x_scale.contiguous() if x_scale is not None else None
    y_scale = y_scale.contiguous() if y_scale is not None else None
    lib.check(x, gy, x_scale, y_scale, out)
    n, c, m, h, wd = geom.n, geom.c, geom.m, geom.oh, geom.ow
    if tuple(x.shape) != (n, c, geom.h, geom.w) or tuple(gy.shape) != (n, m, h, wd):
        raise hip_lib.SaeError("winograd wgrad: x %s, gy %s for a (%d, %d -> %d, %d x %d) layer" % (
            tuple(x.shape), tuple(gy.shape), n, c, m, geom.h, geom.w))
    th, tw = h // 2, wd // 2
    tiles = th * tw
    stream = lib.stream(x)
    dev = x.device
    v = torch.empty((16, n * c, tiles), dtype=torch.float32, device=dev)
    lib.call("wino_input_f32", x.data_ptr(), hip_lib.ptr(x_scale), v.data_ptr(), n * c, geom.h, geom.w, geom.pad, stream)
    e = torch.empty((16, n * m, tiles), dtype=torch.float32, device=dev)
    lib.call("wino_gy_f32", gy.data_ptr(), hip_lib.ptr(y_scale), e.data_ptr(), n * m, h, wd, stream)
    gu = torch.empty((16, m, c), dtype=torch.float32, device=dev)
    n_ws = lib.query("wino_wgrad_gemm_workspace", n, c, m, th, tw)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
    lib.call("wino_wgrad_gemm_f32", v.data_ptr(), e.data_ptr(), gu.data_ptr(), n, c, m, th, tw, ws.data_ptr(), n_ws, stream)
    d = geom.desc()
    if out is None or tuple(out.shape) != tuple(geom.weight_shape()) or not out.is_contiguous():
        out = torch.empty(geom.weight_shape(), dtype=torch.float32, device=dev)
    lib.call("wino_wgrad_output_f32", gu.data_ptr(), out.data_ptr(), m, c, d.w_stride_m, d.w_stride_c, geom.alpha, stream)
    return out
