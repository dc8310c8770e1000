"""The downsampling ResBlock of D / Dpatch as ONE autograd node with a hand-scheduled backward.

Reference semantics: ``(conv2(conv1(x)) + skip(x)) / sqrt(2)`` with conv1 = 3x3 conv + FusedLeakyReLU, conv2 = Blur + 3x3
stride-2 conv + FusedLeakyReLU, skip = Blur + 1x1 stride-2 conv (models/networks/stylegan2_layers.py:672-693 built from
ConvLayer, :612-668).  The forward launches exactly the kernels the module-by-module path launches (bit-identical
output).  What a single node buys is control over the BACKWARD, where autograd's op-by-op schedule pays full HBM passes
for elementwise work between the kernels (profiles/r2_roofline_by_kernel_church256.txt: `bias_act backward` 10.2 ms,
`aten::add_` 3.5 ms, `add_scale backward` 1.2 ms per iteration):

* the backward of conv1's activation (fused_act.py:32-41) rides in the epilogue of the blur's backward
  (sae_upfirdn2d_epilogue_f32): the gradient w.r.t. conv1's output is never written to or re-read from HBM;
* the two gradients of the block input (conv1's data gradient and the skip path's) are summed inside the skip path's
  last kernel (the x2 zero-insert FIR accumulates into conv1's data gradient) instead of by a separate ``add_``;
* the 1/sqrt(2) of the residual merge is folded into the constants of the kernels that consume the output gradient
  (conv2's activation backward, the skip conv's alpha) instead of a pass of its own;
* (forward) the merge itself rides in the skip path's 1x1 conv (sae_conv2d_fwd_residual_f32).

The node stays twice differentiable (the R1 penalty differentiates D and Dpatch twice, swapping_autoencoder_model.py:
143-174): when backward runs with create_graph=True and only the input gradient is wanted (the R1 pattern) it chains the
package's differentiable backward operators on the saved activations -- what autograd does for the module-by-module path,
no recomputation; with weight gradients in the graph it re-expresses the block with the differentiable forward operators
and lets autograd differentiate that.
"""
import math

import torch
from torch.autograd import Function

from .. import hip_lib
from . import conv2d_gemm as cg
from .conv2d_gemm import _Flags, _Geom
from .fused_act import bias_act_bwd_raw
from .upfirdn2d import _adjoint_pad, _flipped, _run as _upfirdn_run, upfirdn2d
from .upsample import add_scale


class ResBlockConfig:
    """Static description of one block: FIR taps (the Blur buffers), pads, equalised-lr scales, activation constants."""
    __slots__ = ("taps2", "pad2", "taps_s", "pad_s", "alpha1", "alpha2", "alpha_s", "slope1", "scale1", "slope2", "scale2",
                 "merge")

    def __init__(self, taps2, pad2, taps_s, pad_s, alpha1, alpha2, alpha_s, slope1, scale1, slope2, scale2, merge):
        self.taps2, self.pad2, self.taps_s, self.pad_s = taps2, tuple(pad2), taps_s, tuple(pad_s)
        self.alpha1, self.alpha2, self.alpha_s = float(alpha1), float(alpha2), float(alpha_s)
        self.slope1, self.scale1, self.slope2, self.scale2 = float(slope1), float(scale1), float(slope2), float(scale2)
        self.merge = float(merge)


class StemConfig:
    """The activated stride-1 ConvLayer in front of a block (D's FromRGB 1x1, Dpatch's 3x3 stem): its output is the block's
    input, so the backward of ITS activation can ride in the kernel that finishes the block's input gradient."""
    __slots__ = ("k", "pad", "alpha", "slope", "scale")

    def __init__(self, k, pad, alpha, slope, scale):
        self.k, self.pad, self.alpha, self.slope, self.scale = int(k), int(pad), float(alpha), float(slope), float(scale)


def stem_composed(x, w0, b0, scfg):
    return cg.conv2d_bias_act(x, w0, b0, stride=1, padding=scfg.pad, alpha=scfg.alpha, negative_slope=scfg.slope,
                              scale=scfg.scale)


def resblock_composed(x, w1, b1, w2, b2, ws, cfg):
    """The block from the differentiable operators (what ResBlock.forward runs module by module)."""
    a1 = cg.conv2d_bias_act(x, w1, b1, stride=1, padding=1, alpha=cfg.alpha1, negative_slope=cfg.slope1, scale=cfg.scale1)
    a1b = upfirdn2d(a1, cfg.taps2, pad=cfg.pad2)
    a2 = cg.conv2d_bias_act(a1b, w2, b2, stride=2, padding=0, alpha=cfg.alpha2, negative_slope=cfg.slope2, scale=cfg.scale2)
    s0 = upfirdn2d(x, cfg.taps_s, down=2, pad=cfg.pad_s)
    s1 = cg.conv2d(s0, ws, None, stride=1, padding=0, alpha=cfg.alpha_s)
    return add_scale(a2, s1, cfg.merge)


def _pad4(pad):
    return (pad[0], pad[1], pad[0], pad[1])


def _k1_epilogue(g, taps, up, pad4, act_ref=None, slope=0.0, scale=1.0, accumulate_into=None):
    """sae_upfirdn2d_epilogue_f32 on an NCHW gradient.  Returns (y, grad_bias or None); `accumulate_into` is updated in
    place and returned as y."""
    lib = hip_lib.get()
    g = g.contiguous()
    lib.check(g, taps, act_ref, accumulate_into)
    n, c, h, w = g.shape
    kh, kw = taps.shape
    oh = h * up + pad4[2] + pad4[3] - kh + 1
    ow = w * up + pad4[0] + pad4[1] - kw + 1
    if accumulate_into is not None:
        if tuple(accumulate_into.shape) != (n, c, oh, ow):
            raise hip_lib.SaeError("k1 epilogue: accumulation target %s, result (%d, %d, %d, %d)" % (
                tuple(accumulate_into.shape), n, c, oh, ow))
        y = accumulate_into
    else:
        y = torch.empty((n, c, oh, ow), dtype=g.dtype, device=g.device)
    gb = ws = None
    n_ws = 0
    if act_ref is not None:
        if tuple(act_ref.shape) != (n, c, oh, ow):
            raise hip_lib.SaeError("k1 epilogue: activation reference %s, result (%d, %d, %d, %d)" % (
                tuple(act_ref.shape), n, c, oh, ow))
        n_ws = lib.query("upfirdn2d_epilogue_workspace", n * c, oh, ow, c, up)
        ws = torch.empty(max(n_ws, 1), dtype=g.dtype, device=g.device)
        gb = torch.empty(c, dtype=g.dtype, device=g.device)
    lib.call("upfirdn2d_epilogue_f32", g.data_ptr(), taps.data_ptr(), y.data_ptr(), n * c, h, w, kh, kw, up,
             pad4[0], pad4[1], pad4[2], pad4[3], hip_lib.ptr(act_ref), float(slope), float(scale), hip_lib.ptr(gb), c,
             1 if accumulate_into is not None else 0, hip_lib.ptr(ws), n_ws, lib.stream(g))
    return y, gb


def _input_grad_graph(ctx, gout, nargs, need):
    """Differentiable input gradient of the block (and stem) from the saved activations: the operators autograd itself
    would chain for the module-by-module path (FusedLeakyReLUFunctionBackward, ConvDataGrad, UpFirDn2dBackward)."""
    from .fused_act import FusedLeakyReLUFunctionBackward
    from .upfirdn2d import UpFirDn2dBackward
    x, w1, b1, w2, b2, ws, a1, a1b, a2, s0, w0, b0, a0 = ctx.saved_tensors
    cfg, scfg = ctx.cfg, ctx.scfg
    g1, g2, gs, g0 = ctx.geoms
    res = [None] * 10
    if not need[0]:
        return tuple(res[:nargs])
    xin = a0 if w0 is not None else x
    g = gout * cfg.merge
    gp2, _ = FusedLeakyReLUFunctionBackward.apply(g, a2, cfg.slope2, cfg.scale2)
    g_a1b = cg.ConvDataGrad.apply(gp2, w2, g2)
    g_a1 = UpFirDn2dBackward.apply(g_a1b, cfg.taps2, (1, 1), (1, 1), _pad4(cfg.pad2), tuple(a1.shape[2:]), tuple(a1b.shape[2:]))
    gp1, _ = FusedLeakyReLUFunctionBackward.apply(g_a1, a1, cfg.slope1, cfg.scale1)
    gin = cg.ConvDataGrad.apply(gp1, w1, g1)
    g_s0 = cg.ConvDataGrad.apply(g, ws, gs)
    gin = gin + UpFirDn2dBackward.apply(g_s0, cfg.taps_s, (1, 1), (2, 2), _pad4(cfg.pad_s), tuple(xin.shape[2:]),
                                        tuple(s0.shape[2:]))
    if w0 is not None:
        gp0, _ = FusedLeakyReLUFunctionBackward.apply(gin, a0, scfg.slope, scfg.scale)
        gin = cg.ConvDataGrad.apply(gp0, w0, g0)
    res[0] = gin
    return tuple(res[:nargs])


class ResBlockFunction(Function):
    """forward(x, w1, b1, w2, b2, ws, cfg[, w0, b0, scfg]): with a stem (w0 given) `x` is the STEM's input and the block
    runs on lrelu(conv(x, w0) + b0) * scale."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, cfg, w0=None, b0=None, scfg=None):
        ctx.set_materialize_grads(False)
        a0 = g0 = None
        xin = x
        if w0 is not None:
            n0, c0, h0, wd0 = x.shape
            g0 = _Geom(n0, c0, h0, wd0, w0.shape[0], scfg.k, 1, scfg.pad, False, scfg.alpha)
            a0 = cg._launch_fused(g0, x, w0, b0, scfg.slope, scfg.scale)
            xin = a0
        n, c, h, w = xin.shape
        m = w2.shape[0]
        g1 = _Geom(n, c, h, w, c, 3, 1, 1, False, cfg.alpha1)
        a1 = cg._launch_fused(g1, xin, w1, b1, cfg.slope1, cfg.scale1)
        a1b = _upfirdn_run(a1, cfg.taps2, (1, 1), (1, 1), _pad4(cfg.pad2))
        g2 = _Geom(n, c, a1b.shape[2], a1b.shape[3], m, 3, 2, 0, False, cfg.alpha2)
        a2 = cg._launch_fused(g2, a1b, w2, b2, cfg.slope2, cfg.scale2)
        s0 = _upfirdn_run(xin, cfg.taps_s, (1, 1), (2, 2), _pad4(cfg.pad_s))
        gs = _Geom(n, c, s0.shape[2], s0.shape[3], m, 1, 1, 0, False, cfg.alpha_s)
        if (gs.n, gs.m, gs.oh, gs.ow) != tuple(a2.shape):
            raise hip_lib.SaeError("ResBlock: branches disagree, %s vs %s" % (tuple(a2.shape), (gs.n, gs.m, gs.oh, gs.ow)))
        # the skip path's 1x1 conv adds the main branch and applies 1 / sqrt(2) on its way out (no separate merge pass,
        # the skip branch's own output never exists in HBM); bit-identical to conv, then add_scale
        out = cg._launch_residual(gs, s0, ws, a2, cfg.merge)
        ctx.cfg, ctx.scfg = cfg, scfg
        ctx.geoms = (g1, g2, gs, g0)
        ctx.has_bias = (b1 is not None, b2 is not None, b0 is not None)
        ctx.save_for_backward(x, w1, b1, w2, b2, ws, a1, a1b, a2, s0, w0, b0, a0)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, w1, b1, w2, b2, ws, a1, a1b, a2, s0, w0, b0, a0 = ctx.saved_tensors
        cfg, scfg = ctx.cfg, ctx.scfg
        nargs = len(ctx.needs_input_grad)                     # 7 without a stem, 10 with one
        need = tuple(ctx.needs_input_grad) + (False,) * (10 - nargs)
        wflag = _Flags.weight_grads
        stem = w0 is not None
        if gout is None:     # undefined = zero: nothing for the activation, explicit zeros for the parameters
            def z(i, t):
                return torch.zeros_like(t) if (t is not None and need[i] and wflag) else None
            return (None, z(1, w1), z(2, b1), z(3, w2), z(4, b2), z(5, ws), None, z(7, w0), z(8, b0), None)[:nargs]
        if torch.is_grad_enabled() and not wflag:
            # create_graph=True inside input_grads_only() -- the R1 penalty's first backward (swapping_autoencoder_model.py:
            # 143-148): only the INPUT gradient is wanted.  It depends on the activations through the leaky-ReLU masks alone
            # (piecewise constant), so the chain of differentiable backward operators on the SAVED activations is exact and
            # nothing has to be recomputed; the graph it records reaches the weights and the incoming gradient.
            return _input_grad_graph(ctx, gout, nargs, need)
        if torch.is_grad_enabled():
            # create_graph=True with weight gradients in the graph: differentiate the composition of differentiable
            # operators (one extra forward of the block).  Neither this package's model nor the drop-in runner reaches this
            # branch -- both evaluate the R1 penalty under input_grads_only() (swapping_autoencoder_model.compute_R1_loss,
            # dropin.wrap_reference_r1 for the reference's own model file); it serves callers that differentiate a block
            # twice without saying that only the input gradient matters
            slots = {0: x, 1: w1, 2: b1, 3: w2, 4: b2, 5: ws, 7: w0, 8: b0}
            wanted = [i for i, t in slots.items() if t is not None and need[i] and (i == 0 or wflag)]
            with torch.enable_grad():
                xin = stem_composed(x, w0, b0, scfg) if stem else x
                out = resblock_composed(xin, w1, b1, w2, b2, ws, cfg)
                grads = torch.autograd.grad(out, [slots[i] for i in wanted], gout, create_graph=True, allow_unused=True)
            res = [None] * 10
            for i, g in zip(wanted, grads):
                res[i] = g
            return tuple(res[:nargs])

        g1, g2, gs, g0 = ctx.geoms
        xin = a0 if stem else x
        gout = gout.contiguous()
        c = cfg.merge
        need_w = [need[i] and wflag for i in range(10)]
        # conv2 branch: activation backward with the merge factor folded into its scale, then dgrad / wgrad
        gp2, gb2 = bias_act_bwd_raw(gout, a2, cfg.slope2, cfg.scale2 * c)
        gw2 = cg.weight_grad(a1b, gp2, g2, w2) if need_w[3] else None
        gb2 = gb2 if (ctx.has_bias[1] and need[4]) else None
        g_a1b = cg._dgrad(gp2, w2, g2)
        del gp2
        # blur backward with conv1's activation backward (and its bias gradient) in the epilogue
        pad2 = _pad4(cfg.pad2)
        adj2 = _adjoint_pad(tuple(a1.shape[2:]), tuple(a1b.shape[2:]), tuple(cfg.taps2.shape), (1, 1), (1, 1), pad2)
        gp1, gb1 = _k1_epilogue(g_a1b, _flipped(cfg.taps2), 1, adj2, act_ref=a1, slope=cfg.slope1, scale=cfg.scale1)
        del g_a1b
        gw1 = cg.weight_grad(xin, gp1, g1, w1) if need_w[1] else None
        gb1 = gb1 if (ctx.has_bias[0] and need[2]) else None
        # skip branch (merge factor folded into alpha); its x2 zero-insert FIR accumulates into conv1's data gradient
        gs_c = _Geom(gs.n, gs.c, gs.h, gs.w, gs.m, gs.k, gs.stride, gs.pad, gs.cm_layout, gs.alpha * c)
        gws = cg.weight_grad(s0, gout, gs_c, ws) if need_w[5] else None
        gx = gw0 = gb0 = None
        need_stem = stem and (need[0] or need_w[7] or (need[8] and wflag))
        if need_stem or (not stem and need[0]):
            gin = cg._dgrad(gp1, w1, g1)
            g_s0 = cg._dgrad(gout, ws, gs_c)
            pad_s = _pad4(cfg.pad_s)
            adj_s = _adjoint_pad(tuple(xin.shape[2:]), tuple(s0.shape[2:]), tuple(cfg.taps_s.shape), (1, 1), (2, 2), pad_s)
            if stem:
                # ... and, when the block's input is the stem's activation, that activation's backward on the same pass:
                # `gin` leaves the kernel as the gradient w.r.t. the stem's PRE-activation, with the stem's bias gradient
                gp0, gb0 = _k1_epilogue(g_s0, _flipped(cfg.taps_s), 2, adj_s, accumulate_into=gin, act_ref=a0, slope=scfg.slope,
                                        scale=scfg.scale)
                gb0 = gb0 if (ctx.has_bias[2] and need[8] and wflag) else None
                gw0 = cg.weight_grad(x, gp0, g0, w0) if need_w[7] else None
                gx = cg._dgrad(gp0, w0, g0) if need[0] else None
            else:
                _k1_epilogue(g_s0, _flipped(cfg.taps_s), 2, adj_s, accumulate_into=gin)
                gx = gin
        return (gx, gw1, gb1, gw2, gb2, gws, None, gw0, gb0, None)[:nargs]


def resblock(x, w1, b1, w2, b2, ws, cfg, stem=None):
    """stem: None or (w0, b0, StemConfig) -- x is then the stem's input."""
    if stem is None:
        return ResBlockFunction.apply(x, w1, b1, w2, b2, ws, cfg)
    return ResBlockFunction.apply(x, w1, b1, w2, b2, ws, cfg, stem[0], stem[1], stem[2])
