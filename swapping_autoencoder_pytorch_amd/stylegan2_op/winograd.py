"""Winograd F(2x2, 3x3) route for the wide 3x3 stride-1 convolutions (pad 1 or 0) (csrc/winograd.hip, include/sae_hip.h: sae_wino_*).

    y = output_transform( 16 x [ conv1x1( input_transform(x)[xi], U[xi] ) ] )        U = weight_transform(w)

The same three C-ABI transforms serve the forward (``F.conv2d`` at models/networks/stylegan2_layers.py:136,315) and the data
gradient (the reversed filter with the channel roles swapped); the 16 transform-domain products are one batched launch of the 1x1
MFMA gather.  ``wgrad`` is the weight gradient on the same sixteen points.  2.25x fewer multiplications for two extra passes over 4x the activation: it pays on layers with many channels on
small maps.  ON by default since round 5 (measured: profiles/r5_winograd_first_session.txt, r5_wino_ab_*.json) for the launches a
per-shape rule selects (route()); where it pays, the ONE-kernel form of csrc/winograd_fused.hip (transforms in registers, no
4x activation round trip) replaces the three-kernel form.  ``SAE_WINOGRAD=0`` turns the route off.
Results differ from the direct kernels' by rounding (~1e-6 relative), so the route is never mixed into bit-identity checks."""
import os

import torch

from .. import hip_lib
from . import weight_prep

FWD, DGRAD, WGRAD = "fwd", "dgrad", "wgrad"
# The one-kernel weight gradient writes one partial gradient per K slice (slices x Mp x 9 x Cp floats: 226 MB for a 512 -> 512
# layer cut into 24 slices) into a slab taken from torch's caching allocator for the duration of the call.  Layers whose slab
# would exceed this take the three-kernel form instead (none of the three presets' layers comes near it).
FUSED_WGRAD_SLAB_BYTES = 1 << 30


class _Config:
    """Read ONCE (import, or configure()): the conv wrappers ask eligible() on every call, which must not cost an environment
    lookup.  SAE_WINOGRAD=0 turns the route off; SAE_WINOGRAD_MIN_C overrides the smallest channel count (and with it the
    measured per-shape rule: every 3x3 stride-1 layer of at least that many channels then takes the route -- the tests'
    setting); SAE_WINOGRAD_FUSED=0 keeps the three-kernel form everywhere."""

    def __init__(self):
        self.enabled = os.environ.get("SAE_WINOGRAD", "1") != "0"
        mc = os.environ.get("SAE_WINOGRAD_MIN_C")
        self.min_c = int(mc) if mc is not None else None
        self.fused = os.environ.get("SAE_WINOGRAD_FUSED", "1") != "0"
        self.memo = {}


_CFG = _Config()


def configure(enabled=None, min_c="keep", fused=None):
    """Change the route's switches at run time (tests, bench A/B); returns the previous (enabled, min_c, fused)."""
    prev = (_CFG.enabled, _CFG.min_c, _CFG.fused)
    if enabled is not None:
        _CFG.enabled = bool(enabled)
    if min_c != "keep":
        _CFG.min_c = min_c
    if fused is not None:
        _CFG.fused = bool(fused)
    _CFG.memo.clear()
    return prev


class override:
    """with winograd.override(enabled=True, min_c=8): ...  -- configure() for the duration of a block"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.prev = configure(**self.kw)
        return self

    def __exit__(self, *exc):
        configure(*self.prev)
        return False


def enabled():
    return _CFG.enabled


def _pow2_ceil_log2(v):
    e = 0
    while (1 << e) < v:
        e += 1
    return e


def _fused_workgroups(n, tiles_h, tiles_w, m):
    """Workgroups sae_wino_fused_conv_f32 launches: 64-tile blocks (BN images x BH x BW tiles, csrc/winograd_fused.hip's
    decomposition) x 64-channel blocks."""
    bw = min(_pow2_ceil_log2(tiles_w), 4)
    bh = min(_pow2_ceil_log2(tiles_h), 6 - bw)
    bn = 64 >> (bw + bh)
    blocks = -(-tiles_w // (1 << bw)) * -(-tiles_h // (1 << bh)) * -(-n // bn)
    return blocks * -(-m // 64)


def _route_of(geom, op):
    """None (direct kernels), "fused" (one kernel) or "unfused" (transform / sixteen products / transform) for a 3x3 stride-1
    layer: the cheapest of three estimates in milliseconds, each fitted to the same-box A/B of every such launch of the church256
    iteration (tools/wino_ab.py -> profiles/r5_wino_ab_church256_fused.json; the three-kernel rule alone keeps 27.5 of the 27.7 ms
    per iteration that picking the faster of direct / three-kernel per launch would save):
      direct    FLOPs at 130 TFLOP/s (the gathers' plateau), never under 35 us
      unfused   eligible from 256 channels, 16 tiles per image and ~1000 tiles in all (weight gradient: n * tiles * channels >=
                4M): 0.62 / 0.66 / 0.78 of the direct estimate at >= 512 / 384 / 256 channels (measured 0.60 - 0.80)
      fused     rounds of 256 workgroups (one per CU: 256 accumulators per lane) x (0.162 ms x C / 512 + 6 us) per round --
                within 5 % of every measured row; forward and data gradient only"""
    if geom.k == 3 and geom.stride == 2:
        return _route_of_stride2(geom, op)
    if geom.k != 3 or geom.stride != 1 or geom.pad not in (0, 1) or (geom.h & 1) or (geom.w & 1):
        return None
    cmin = min(geom.c, geom.m)
    tiles = (geom.oh >> 1) * (geom.ow >> 1)
    # the one-kernel forms address their activations with 32-bit byte offsets
    fused_ok = _CFG.fused and max(geom.n * geom.c * geom.h * geom.w, geom.n * geom.m * geom.oh * geom.ow) * 4 < (1 << 31)
    if _CFG.min_c is not None:                        # explicit threshold (tests): everything at least that wide
        if cmin < _CFG.min_c:
            return None
        if op != WGRAD and fused_ok and (geom.ow if op == DGRAD else geom.w) >= 4:
            return "fused"
        if op == WGRAD and fused_ok and geom.ow % 16 == 0:
            return "fused"
        return "unfused"
    if op == WGRAD:
        # three-kernel form (two transforms + sixteen K-sliced 1x1 weight gradients): from 256 channels on enough pixels;
        # one-kernel form (sae_wino_fused_wgrad_f32): output rows a multiple of 16 pixels, enough 8-tile chunks for whole
        # rounds of 64 x 64-channel workgroups with slices of >= 8 chunks
        if fused_ok and geom.ow % 16 == 0 and cmin >= 64 and geom.n * tiles >= 4096:
            return "fused"
        return "unfused" if (cmin >= 256 and geom.n * tiles * cmin >= (4 << 20)) else None
    direct = max(2.0 * geom.n * geom.m * geom.oh * geom.ow * geom.c * 9 / 130e9, 0.035)
    best, cost = None, direct
    if cmin >= 256 and tiles >= 16 and geom.n * tiles >= 1024:
        est = direct * (0.62 if cmin >= 512 else 0.66 if cmin >= 384 else 0.78)
        if est < cost:
            best, cost = "unfused", est
    if fused_ok:
        # the product that is computed: forward c -> m on the oh x ow grid; data gradient m -> c on the h x w grid
        cin, cout, th, tw, in_w = ((geom.m, geom.c, geom.h >> 1, geom.w >> 1, geom.ow) if op == DGRAD else
                                   (geom.c, geom.m, geom.oh >> 1, geom.ow >> 1, geom.w))
        # (from 64 channels and ~2000 tiles: below that both routes are launch-bound, the estimates mean nothing, and the
        # small presets -- incl. the golden micro steps of the parity tests -- stay on the direct kernels)
        if in_w >= 4 and cmin >= 64 and geom.n * th * tw >= 2048:
            wgs = _fused_workgroups(geom.n, th, tw, cout)
            est = -(-wgs // 256) * (0.162 * (-(-cin // 8) * 8) / 512.0 + 0.006)
            if est < cost:
                best, cost = "fused", est
    return best


def _route_of_stride2(geom, op):
    """None or "s2poly": the data gradient / transposed convolution of a 3x3 stride-2 pad-0 layer between a (2h+1) x (2w+1) map and
    an h x w map on its polyphase minimal-filtering form (csrc/s2wino.hip: 25 instead of 36 multiplications per 2x2 of small-side
    positions).  Same-box A/B of the step's launches (tools/ab_s2wino.py -> profiles/r6_ab_s2wino_dgrad.txt): 1.1 - 1.2x the direct
    kernel on the small-side maps of 8 .. 32 with >= 256 contraction channels (the direct kernel's tiles quantise badly on the
    17 / 33 / 65-wide grids), behind it on the 64 / 128 maps -- which stay on the direct kernel."""
    if op != DGRAD or geom.pad != 0 or geom.h != 2 * geom.oh + 1 or geom.w != 2 * geom.ow + 1:
        return None
    if (geom.oh & 1) or (geom.ow & 1) or geom.ow < 4 or geom.n * geom.m * geom.oh * geom.ow * 4 >= (1 << 31):
        return None
    if _CFG.min_c is not None:                        # explicit threshold (tests): everything at least that wide
        return "s2poly" if min(geom.c, geom.m) >= _CFG.min_c else None
    if geom.m >= 256 and geom.c >= 128 and 8 <= geom.oh <= 32 and 8 <= geom.ow <= 32 and geom.n * geom.oh * geom.ow >= 4096:
        return "s2poly"
    return None


def route(geom, op=FWD):
    if not _CFG.enabled:
        return None
    key = (geom.key, op)
    r = _CFG.memo.get(key, 0)
    if r == 0:
        r = _CFG.memo[key] = _route_of(geom, op)
    if r is not None and hip_lib.get().query("get_conv_math") != 0:      # exact-fp32 arithmetic only
        return None
    return r


def eligible(geom, op=FWD):
    return route(geom, op) is not None


def _weights(lib, w, kind, cout, cin, sm, sc, flip, alpha, row_scale, col_scale, tag):
    """Transform-domain weights of `w`, kept per parameter version (weight_prep.cached) when the factors are a function of the
    parameter alone (tag: () = no factors, a hashable description, or None = unknown -> rebuilt per call)."""
    def build():
        if kind == "s2poly":
            u = torch.empty(lib.query("s2wino_weights_floats", cout, cin), dtype=torch.float32, device=w.device)
            lib.call("s2wino_weights_f32", w.data_ptr(), hip_lib.ptr(row_scale), hip_lib.ptr(col_scale), u.data_ptr(), cout, cin,
                     sm, sc, flip, alpha, lib.stream(w))
            return u
        if kind == "fused":
            u = torch.empty(lib.query("wino_fused_weights_floats", cout, cin), dtype=torch.float32, device=w.device)
            lib.call("wino_fused_weights_f32", w.data_ptr(), hip_lib.ptr(row_scale), hip_lib.ptr(col_scale), u.data_ptr(), cout, cin,
                     sm, sc, flip, alpha, lib.stream(w))
        else:
            u = torch.empty((16, cout, cin), dtype=torch.float32, device=w.device)
            lib.call("wino_weights_f32", w.data_ptr(), hip_lib.ptr(row_scale), hip_lib.ptr(col_scale), u.data_ptr(), cout, cin, sm,
                     sc, flip, alpha, lib.stream(w))
        return u
    if row_scale is None and col_scale is None:
        tag = ()
    if tag is None:
        return build()
    return weight_prep.cached(w, ("wino", kind, flip, float(alpha), tag), build)


def conv(x, w, geom, transpose=False, bias=None, act=None, x_scale=None, row_scale=None, col_scale=None, out_scale=None,
         noise=None, noise_weight=None, factor_tag=None, kind=None):
    """alpha * conv(x * x_scale, w') (transpose=False: x is [n, c, h, w]) or alpha * conv^T(gy * x_scale, w') (transpose=True: x
    is the output gradient [n, m, h, w]) with w' = w * row_scale * col_scale along the OUTPUT / CONTRACTION axes of the product
    that is computed; times out_scale [n, outputs] if given; then, with act = (slope, scale),
    lrelu((. + noise_weight * noise[n]) + bias) * scale (noise, bias optional).  factor_tag: what row_scale / col_scale are as a
    function of the weight parameter alone (weight_prep), None = unknown.  kind: "fused" / "unfused" (default: route())."""
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
    if kind is None:
        kind = route(geom, DGRAD if transpose else FWD) or "unfused"
    stream = lib.stream(x)
    dev = x.device
    if kind == "s2poly":        # stride-2 data gradient: x is the small side [n, m, oh, ow], the result the (2 oh + 1) x (2 ow + 1) side
        if not transpose or act is not None or noise is not None or bias is not None:
            raise hip_lib.SaeError("winograd conv: the polyphase stride-2 form covers the data gradient / transposed convolution only")
        if tuple(x.shape) != (geom.n, geom.m, geom.oh, geom.ow):
            raise hip_lib.SaeError("polyphase stride-2 data gradient: input %s, expected (%d, %d, %d, %d)" % (
                tuple(x.shape), geom.n, geom.m, geom.oh, geom.ow))
        u = _weights(lib, w, kind, geom.c, geom.m, d.w_stride_c, d.w_stride_m, 1, geom.alpha, row_scale, col_scale, factor_tag)
        y = torch.empty((geom.n, geom.c, geom.h, geom.w), dtype=torch.float32, device=dev)
        lib.call("s2wino_dgrad_f32", x.data_ptr(), hip_lib.ptr(x_scale), u.data_ptr(), hip_lib.ptr(out_scale), y.data_ptr(), geom.n,
                 geom.m, geom.c, geom.oh, geom.ow, stream)
        return y
    u = _weights(lib, w, kind, cout, cin, sm, sc, flip, geom.alpha, row_scale, col_scale, factor_tag)
    y = torch.empty((n, cout, h, wd), dtype=torch.float32, device=dev)
    slope, scale = act if act is not None else (0.0, 1.0)
    if kind == "fused":
        lib.call("wino_fused_conv_f32", x.data_ptr(), hip_lib.ptr(x_scale), u.data_ptr(), hip_lib.ptr(out_scale), hip_lib.ptr(noise),
                 hip_lib.ptr(noise_weight), hip_lib.ptr(bias), y.data_ptr(), n, cin, cout, ih, iw, pad, 1 if act is not None else 0,
                 float(slope), float(scale), stream)
        return y
    th, tw = h // 2, wd // 2
    tiles = th * tw
    v = torch.empty((16, n * cin, tiles), dtype=torch.float32, device=dev)
    lib.call("wino_input_f32", x.data_ptr(), hip_lib.ptr(x_scale), v.data_ptr(), n * cin, ih, iw, pad, stream)
    md = torch.empty((16, n * cout, tiles), dtype=torch.float32, device=dev)
    n_ws = lib.query("wino_gemm_workspace", n, cin, cout, th, tw)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
    lib.call("wino_gemm_f32", v.data_ptr(), u.data_ptr(), md.data_ptr(), n, cin, cout, th, tw, ws.data_ptr(), n_ws, stream)
    lib.call("wino_output_f32", md.data_ptr(), hip_lib.ptr(out_scale), hip_lib.ptr(noise), hip_lib.ptr(noise_weight),
             hip_lib.ptr(bias), y.data_ptr(), n * cout, cout, h, wd, 1 if act is not None else 0, float(slope), float(scale), stream)
    return y


def wgrad(x, gy, geom, out=None, x_scale=None, y_scale=None, kind=None):
    """alpha * sum over images and pixels of (gy * y_scale) (x) (x * x_scale) as the layer's weight gradient (geom.weight_shape()),
    on the sixteen points: gw = G^T [ sum (A e A^T) o (B^T d B) ] G.  out: an existing tensor of that shape to write into (a
    slot of an armed gradient bucket).  kind: "fused" (one kernel + the slice reduction) / "unfused" (default: route())."""
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
    if kind is None:
        kind = route(geom, WGRAD) or "unfused"
    if kind == "fused" and (gy.data_ptr() & 15):          # the kernel fetches gy tile pairs as aligned 16-byte loads
        kind = "unfused"
    n_ws = lib.query("wino_fused_wgrad_workspace", n, c, m, geom.h, geom.w, geom.pad) if kind == "fused" else 0
    if n_ws * 4 > FUSED_WGRAD_SLAB_BYTES:                 # slices x Mp x 9 x Cp partial gradients: capped, not open-ended
        kind = "unfused"
    if kind == "fused":
        d = geom.desc()
        if out is None or tuple(out.shape) != tuple(geom.weight_shape()) or not out.is_contiguous():
            out = torch.empty(geom.weight_shape(), dtype=torch.float32, device=x.device)
        ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=x.device)
        lib.call("wino_fused_wgrad_f32", x.data_ptr(), hip_lib.ptr(x_scale), gy.data_ptr(), hip_lib.ptr(y_scale), out.data_ptr(), n, c,
                 m, geom.h, geom.w, geom.pad, d.w_stride_m, d.w_stride_c, geom.alpha, ws.data_ptr(), n_ws, lib.stream(x))
        return out
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
