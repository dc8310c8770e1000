"""Dense conv2d / conv_transpose2d / linear on the gfx950 fp32-MFMA kernels.

These replace the ATen calls of the reference's layer library — ``F.conv2d``
(models/networks/stylegan2_layers.py:136,175,182,315,321), ``F.conv_transpose2d`` (:306) and
``F.linear`` (:177,186) — with the same argument meaning.  The equalised-learning-rate factor
``weight * scale`` the reference recomputes as a separate elementwise kernel on every call
(:138,175-187,275,285) is folded into the kernels as ``alpha``.

Differentiation: three primitives (forward, dgrad, wgrad) are closed under differentiation —
the derivative of each is expressed with the other two — so gradients of any order exist.  D and
Dpatch need second order for the R1 penalty (swapping_autoencoder_model.py:143-174).
``ctx.needs_input_grad`` is honoured so the forward-only E/G passes of a D step and the
dgrad-only D/Dpatch passes of a G step (SURVEY.md §7 "grad-mode awareness") launch nothing they
do not need.
"""
import ctypes as C
import threading

import torch
from torch.autograd import Function

from .. import hip_lib
from ..hip_lib import SAE_CONV_DGRAD, SAE_CONV_FWD, SAE_CONV_WGRAD, ConvDesc
from . import weight_prep, winograd


_WINO_OP = {SAE_CONV_FWD: winograd.FWD, SAE_CONV_DGRAD: winograd.DGRAD, SAE_CONV_WGRAD: winograd.WGRAD}


class _Flags:
    weight_grads = True
    depth = 0                      # nesting / concurrency count of input_grads_only (weight_grads == (depth == 0))
    lock = threading.Lock()


class input_grads_only:
    """Context manager: inside it, backward passes skip the WEIGHT gradients of convs / linears.

    ``torch.autograd.grad(out, inputs=[image])`` prunes the weight-gradient branches of built-in
    ops, but a custom ``Function`` only sees the static ``ctx.needs_input_grad`` and would launch a
    wgrad kernel per layer whose result nobody reads.  The R1 penalty's first backward
    (swapping_autoencoder_model.py:143-148,169-174) is exactly that case."""

    def __enter__(self):
        # (a count, not save / restore: backward passes read the flag from autograd's device threads, so it has to be
        # process-wide, and entries that interleave across threads -- nn.DataParallel replicas of a host program -- must
        # leave it True once the last one has exited)
        with _Flags.lock:
            _Flags.depth += 1
            _Flags.weight_grads = False

    def __exit__(self, *exc):
        with _Flags.lock:
            _Flags.depth -= 1
            _Flags.weight_grads = _Flags.depth == 0
        return False


class _Geom:
    """Immutable description of one conv problem (forward orientation) + weight layout."""
    __slots__ = ("n", "c", "h", "w", "m", "k", "stride", "pad", "oh", "ow", "cm_layout", "alpha", "key")

    def __init__(self, n, c, h, w, m, k, stride, pad, cm_layout, alpha):
        self.n, self.c, self.h, self.w, self.m, self.k = n, c, h, w, m, k
        self.stride, self.pad, self.cm_layout, self.alpha = stride, pad, cm_layout, float(alpha)
        self.oh = (h + 2 * pad - k) // stride + 1
        self.ow = (w + 2 * pad - k) // stride + 1
        self.key = (n, c, h, w, m, k, stride, pad, bool(cm_layout))      # what the descriptor is built from (weight_prep's memo)
        if self.oh < 1 or self.ow < 1:
            raise hip_lib.SaeError("conv2d: empty output for input %dx%d k=%d stride=%d pad=%d" % (h, w, k, stride, pad))

    def weight_shape(self):
        return (self.c, self.m, self.k, self.k) if self.cm_layout else (self.m, self.c, self.k, self.k)

    def desc(self):
        d = ConvDesc()
        d.n, d.c, d.h, d.w, d.m, d.oh, d.ow = self.n, self.c, self.h, self.w, self.m, self.oh, self.ow
        d.kh = d.kw = self.k
        d.stride, d.pad = self.stride, self.pad
        kk = self.k * self.k
        if self.cm_layout:
            d.w_stride_m, d.w_stride_c = kk, self.m * kk
        else:
            d.w_stride_m, d.w_stride_c = self.c * kk, kk
        return d


def _launch(name, op, geom, a, b, out_shape, out=None):
    lib = hip_lib.get()
    a = a.contiguous()
    b = b.contiguous()
    lib.check(a, b, out)
    d = geom.desc()
    n_ws = lib.query("conv2d_workspace", C.byref(d), op)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=a.device)
    if out is None or tuple(out.shape) != tuple(out_shape):
        out = torch.empty(out_shape, dtype=torch.float32, device=a.device)
    prepared = weight_prep.attach(lib, d, None, op, b, geom.alpha, gkey=geom.key) if op != SAE_CONV_WGRAD else None   # noqa: F841 (kept alive)
    lib.call(name, a.data_ptr(), b.data_ptr(), out.data_ptr(), C.byref(d), geom.alpha, ws.data_ptr(), n_ws,
             lib.stream(a))
    return out


def _launch_fused(geom, x, w, bias, slope, scale):
    if winograd.eligible(geom, winograd.FWD):
        return winograd.conv(x, w, geom, bias=bias, act=(slope, scale))
    lib = hip_lib.get()
    x = x.contiguous()
    w = w.contiguous()
    bias = bias.contiguous() if bias is not None else None
    lib.check(x, w, bias)
    d = geom.desc()
    n_ws = lib.query("conv2d_workspace", C.byref(d), SAE_CONV_FWD)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=x.device)
    out = torch.empty((geom.n, geom.m, geom.oh, geom.ow), dtype=torch.float32, device=x.device)
    prepared = weight_prep.attach(lib, d, None, SAE_CONV_FWD, w, geom.alpha, gkey=geom.key)   # noqa: F841
    lib.call("conv2d_fwd_bias_act_f32", x.data_ptr(), w.data_ptr(), hip_lib.ptr(bias), out.data_ptr(), C.byref(d),
             geom.alpha, float(slope), float(scale), ws.data_ptr(), n_ws, lib.stream(x))
    return out


def _launch_residual(geom, x, w, residual, res_scale):
    """(alpha * conv(x, w) + residual) * res_scale in one kernel (sae_conv2d_fwd_residual_f32)."""
    lib = hip_lib.get()
    x = x.contiguous()
    w = w.contiguous()
    residual = residual.contiguous()
    lib.check(x, w, residual)
    if tuple(residual.shape) != (geom.n, geom.m, geom.oh, geom.ow):
        raise hip_lib.SaeError("conv + residual: residual %s, output (%d, %d, %d, %d)" % (
            tuple(residual.shape), geom.n, geom.m, geom.oh, geom.ow))
    d = geom.desc()
    n_ws = lib.query("conv2d_workspace", C.byref(d), SAE_CONV_FWD)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=x.device)
    out = torch.empty_like(residual)
    prepared = weight_prep.attach(lib, d, None, SAE_CONV_FWD, w, geom.alpha, gkey=geom.key)   # noqa: F841
    lib.call("conv2d_fwd_residual_f32", x.data_ptr(), w.data_ptr(), residual.data_ptr(), out.data_ptr(), C.byref(d), geom.alpha,
             float(res_scale), ws.data_ptr(), n_ws, lib.stream(x))
    return out


def _fwd(x, w, g):
    if winograd.eligible(g, winograd.FWD):
        return winograd.conv(x, w, g)
    return _launch("conv2d_fwd_f32", SAE_CONV_FWD, g, x, w, (g.n, g.m, g.oh, g.ow))


def _dgrad(gy, w, g):
    if winograd.eligible(g, winograd.DGRAD):
        return winograd.conv(gy, w, g, transpose=True)
    return _launch("conv2d_dgrad_f32", SAE_CONV_DGRAD, g, gy, w, (g.n, g.c, g.h, g.w))


def _wgrad(x, gy, g, out=None):
    if winograd.eligible(g, winograd.WGRAD):
        return winograd.wgrad(x, gy, g, out=out)
    return _launch("conv2d_wgrad_f32", SAE_CONV_WGRAD, g, x, gy, g.weight_shape(), out=out)


def weight_grad(x, gy, geom, weight):
    """The weight gradient inside a backward pass: a differentiable node when the pass builds a graph (create_graph=True);
    otherwise the bare kernel, writing straight into the parameter's slot of an armed gradient bucket when there is one
    (grad_allreduce.claim_destination: the all-reduce then needs no gradient -> bucket copy)."""
    if torch.is_grad_enabled():
        return ConvWeightGrad.apply(x, gy, geom)
    from ..grad_allreduce import claim_destination
    return _wgrad(x, gy, geom, out=claim_destination(weight))


def _zero_param_grad(ctx, index, param):
    """Explicit zero gradient of a weight-like input when the output gradient is undefined."""
    if ctx.needs_input_grad[index] and _Flags.weight_grads:
        return torch.zeros_like(param)
    return None


class ConvForward(Function):
    """y = alpha * conv(x, w)"""

    @staticmethod
    def forward(ctx, x, w, geom):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        ctx.geom = geom
        ctx.save_for_backward(x, w)
        return _fwd(x, w, geom)

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            # an undefined output gradient means "zero": nothing for the activation, an explicit zero for the
            # parameter (what a materialised zero gradient would have produced, without the kernels)
            return None, _zero_param_grad(ctx, 1, ctx.saved_tensors[1]), None
        x, w = ctx.saved_tensors
        gx = ConvDataGrad.apply(gy, w, ctx.geom) if ctx.needs_input_grad[0] else None
        gw = weight_grad(x, gy, ctx.geom, w) if (ctx.needs_input_grad[1] and _Flags.weight_grads) else None
        return gx, gw, None


class ConvDataGrad(Function):
    """gx = alpha * conv^T(gy, w)   (also the forward of the stride-2 transposed conv)"""

    @staticmethod
    def forward(ctx, gy, w, geom):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        ctx.geom = geom
        ctx.save_for_backward(gy, w)
        return _dgrad(gy, w, geom)

    @staticmethod
    def backward(ctx, ggx):
        if ggx is None:
            return None, _zero_param_grad(ctx, 1, ctx.saved_tensors[1]), None
        gy, w = ctx.saved_tensors
        g_gy = ConvForward.apply(ggx, w, ctx.geom) if ctx.needs_input_grad[0] else None
        g_w = ConvWeightGrad.apply(ggx, gy, ctx.geom) if ctx.needs_input_grad[1] else None
        return g_gy, g_w, None


class ConvWeightGrad(Function):
    """gw = alpha * sum_pixels gy (x) x"""

    @staticmethod
    def forward(ctx, x, gy, geom):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        ctx.geom = geom
        ctx.save_for_backward(x, gy)
        return _wgrad(x, gy, geom)

    @staticmethod
    def backward(ctx, ggw):
        if ggw is None:
            return None, None, None
        x, gy = ctx.saved_tensors
        g_x = ConvDataGrad.apply(gy, ggw, ctx.geom) if ctx.needs_input_grad[0] else None
        g_gy = ConvForward.apply(x, ggw, ctx.geom) if ctx.needs_input_grad[1] else None
        return g_x, g_gy, None


class ConvBiasAct(Function):
    """lrelu(alpha * conv(x, w) + bias) * scale in ONE kernel (the activation rides in the MFMA
    kernel's epilogue): replaces the Conv -> FusedLeakyReLU pair of ConvLayer
    (stylegan2_layers.py:642-659) and saves a full read + write of the activation tensor.  The
    backward is the reference's: the activation gradient from the saved OUTPUT (fused_act.py:23-41),
    then dgrad / wgrad of the conv — all differentiable again."""

    @staticmethod
    def forward(ctx, x, w, bias, geom, slope, scale):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        out = _launch_fused(geom, x, w, bias, slope, scale)
        ctx.cfg = (geom, slope, scale, bias is not None)
        ctx.save_for_backward(x, w, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            w = ctx.saved_tensors[1]
            gb = torch.zeros(w.shape[0], dtype=w.dtype, device=w.device) if (ctx.cfg[3] and ctx.needs_input_grad[2]) else None
            return None, _zero_param_grad(ctx, 1, w), gb, None, None, None
        from .fused_act import FusedLeakyReLUFunctionBackward
        x, w, out = ctx.saved_tensors
        geom, slope, scale, has_bias = ctx.cfg
        g_pre, g_bias = FusedLeakyReLUFunctionBackward.apply(gout, out, slope, scale)
        gx = ConvDataGrad.apply(g_pre, w, geom) if ctx.needs_input_grad[0] else None
        gw = weight_grad(x, g_pre, geom, w) if (ctx.needs_input_grad[1] and _Flags.weight_grads) else None
        gb = g_bias if (has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None, None, None


# ------------------------------------------------------------------------------------------------
# style-modulated convolution
# ------------------------------------------------------------------------------------------------
def _launch_mod(name, op, geom, a, b, out_shape, x_scale=None, y_scale=None, wm_scale=None, wc_scale=None, factor_tag=None):
    """One sae_modconv2d_* call: the plain operation `op` on a * factor, b (weights or second activation) with the
    optional [N, C] activation factors and per-channel weight factors staged inside the kernels.  factor_tag: what the weight
    factors are as a function of the weight parameter alone (weight_prep.attach), None = unknown (no prepared weights)."""
    if winograd.eligible(geom, _WINO_OP[op]):
        # the factors of sae_conv2d_mod in the route's terms: the activation factor of the operation's input; the weight factors by
        # the axes of the product that is computed (the data gradient's outputs are the c axis)
        if op == SAE_CONV_WGRAD:      # a = x, b = gy
            return winograd.wgrad(a, b, geom, x_scale=x_scale, y_scale=y_scale)
        if op == SAE_CONV_FWD:
            return winograd.conv(a, b, geom, x_scale=x_scale, row_scale=wm_scale, col_scale=wc_scale, factor_tag=factor_tag)
        return winograd.conv(a, b, geom, transpose=True, x_scale=y_scale, row_scale=wc_scale, col_scale=wm_scale,
                             factor_tag=factor_tag)
    lib = hip_lib.get()
    a = a.contiguous()
    b = b.contiguous()
    factors = [None if f is None else f.contiguous() for f in (x_scale, y_scale, wm_scale, wc_scale)]
    lib.check(a, b, *factors)
    d = geom.desc()
    mod = hip_lib.ConvMod(*[hip_lib.ptr(f) for f in factors])
    n_ws = lib.query("conv2d_workspace", C.byref(d), op)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=a.device)
    out = torch.empty(out_shape, dtype=torch.float32, device=a.device)
    if op != SAE_CONV_WGRAD:
        tag = () if (wm_scale is None and wc_scale is None) else factor_tag
        prepared = weight_prep.attach(lib, d, mod, op, b, geom.alpha, tag, gkey=geom.key)   # noqa: F841
    lib.call(name, a.data_ptr(), b.data_ptr(), out.data_ptr(), C.byref(d), C.byref(mod), geom.alpha, ws.data_ptr(), n_ws,
             lib.stream(a))
    return out


def _weight_demod(w, alpha, eps):
    """rsqrt(sum_{i,ky,kx} (alpha w)^2 + eps) per output channel (stylegan2_layers.py:290-292) in one launch"""
    lib = hip_lib.get()
    w = w.contiguous()
    lib.check(w)
    d = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
    lib.call("weight_demod_f32", w.data_ptr(), d.data_ptr(), w.shape[0], w[0].numel(), float(alpha), float(eps), lib.stream(w))
    return d


def _weight_demod_bwd(geff, w, d, alpha, out=None):
    """The parameter's gradient from the effective-weight gradient of a conv that ran on alpha * d[o] * w."""
    lib = hip_lib.get()
    geff, w = geff.contiguous(), w.contiguous()
    lib.check(geff, w, d)
    gw = torch.empty_like(w) if out is None else out
    lib.call("weight_demod_bwd_f32", geff.data_ptr(), w.data_ptr(), d.data_ptr(), gw.data_ptr(), w.shape[0], w[0].numel(),
             float(alpha), lib.stream(w))
    return gw


class ModulatedConv(Function):
    """ModulatedConv2d's arithmetic (stylegan2_layers.py:280-321) as one kernel per operation:

        out = conv(x * s[:, :, None, None],  W * alpha * demod[:, None, None, None])          (plain, stride 1)
        out = conv_transpose(x * s[:, :, None, None],  W * alpha * demod ..., stride 2)        (upsampling form)

    The style factor `s` [N, I] is applied to the activation while the conv kernel stages it into LDS and the
    demodulation factor `demod` [O] while it re-lays the weight, so neither x * s nor the modulated weight is ever
    written to HBM (the reference materialises both, plus `batch` copies of the weight, :287).  `w` is the parameter's
    [O, I, k, k] slice in both forms.  Backward: the data gradient kernel (demod folded), then ONE pass that forms
    grad_x = g * s and grad_s = sum_hw g * x, and the weight-gradient kernel with `s` folded into its operand staging;
    the gradients w.r.t. `w` and `demod` follow from the effective-weight gradient by the product rule.  First-order
    only (the generator is never differentiated twice; the R1 penalties touch D and Dpatch)."""

    @staticmethod
    def forward(ctx, x, s, w, demod, geom, transposed, demod_eps=None, demod_alpha=1.0):
        ctx.set_materialize_grads(False)
        ctx.cfg = (geom, transposed)
        ctx.own_demod = demod_eps is not None
        ctx.demod_alpha = demod_alpha
        if ctx.own_demod:
            # the demodulation factor d = rsqrt(sum (demod_alpha w)^2 + eps) is part of this node: one kernel here, and one
            # in the backward that turns the gradient of d * w into the parameter's (csrc/modulate.hip weight_demod_*;
            # ~20 ATen launches per conv and pass otherwise).  `demod` must be None.
            demod = _weight_demod(w, demod_alpha, demod_eps)
        ctx.save_for_backward(x, s, w, demod)
        # the demodulation factor formed inside this node is a function of the weight alone: prepared weights may be shared
        ctx.factor_tag = tag = ("demod", float(demod_alpha), float(demod_eps)) if ctx.own_demod else None
        if transposed:      # dgrad of the forward-orientation problem: x is its y side, the result its x side
            return _launch_mod("modconv2d_dgrad_f32", SAE_CONV_DGRAD, geom, x, w, (geom.n, geom.c, geom.h, geom.w),
                               y_scale=s, wc_scale=demod, factor_tag=tag)
        return _launch_mod("modconv2d_fwd_f32", SAE_CONV_FWD, geom, x, w, (geom.n, geom.m, geom.oh, geom.ow),
                           x_scale=s, wm_scale=demod, factor_tag=tag)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, s, w, demod = ctx.saved_tensors
        geom, transposed = ctx.cfg
        if gout is None:
            gw = torch.zeros_like(w) if (ctx.needs_input_grad[2] and _Flags.weight_grads) else None
            gd = torch.zeros_like(demod) if (demod is not None and not ctx.own_demod and ctx.needs_input_grad[3]) else None
            return None, None, gw, gd, None, None, None, None
        gx, gs, gw, gd = _modconv_backward(ctx, gout, x, s, w, demod, geom, transposed)
        return gx, gs, gw, gd, None, None, None, None


def _modconv_backward(ctx, gout, x, s, w, demod, geom, transposed, in_ticket=None):
    """(grad x, grad s, grad w, grad demod) of a ModulatedConv-type node; ctx: needs_input_grad[0..3] = x, s, w, demod,
    own_demod, demod_alpha"""
    gout = gout.contiguous()
    gx = gs = gw = gd = None
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
        tag = getattr(ctx, "factor_tag", None)
        if transposed:
            g = _launch_mod("modconv2d_fwd_f32", SAE_CONV_FWD, geom, gout, w, (geom.n, geom.m, geom.oh, geom.ow),
                            wc_scale=demod, factor_tag=tag)
        else:
            g = _launch_mod("modconv2d_dgrad_f32", SAE_CONV_DGRAD, geom, gout, w, (geom.n, geom.c, geom.h, geom.w),
                            wm_scale=demod, factor_tag=tag)
        from .modulate import fusable, plane_scale_backward, plane_scale_dot_act
        if in_ticket is not None and fusable(g) and ctx.needs_input_grad[0]:
            gx, gs = plane_scale_dot_act(g, x, s, in_ticket)    # ... and the producer's activation backward (ActTicket)
        elif fusable(g):
            gx, gs = plane_scale_backward(g, x, s)       # g * s and sum_hw g * x in one pass
        else:
            gx, gs = g * s[:, :, None, None], (g * x).sum(dim=(2, 3))
    if (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) and _Flags.weight_grads:
        if transposed:
            geff = _launch_mod("modconv2d_wgrad_f32", SAE_CONV_WGRAD, geom, gout, x, geom.weight_shape(), y_scale=s)
        else:
            geff = _launch_mod("modconv2d_wgrad_f32", SAE_CONV_WGRAD, geom, x, gout, geom.weight_shape(), x_scale=s)
        # geff = d(loss)/d(demod[o] * w) (the kernels' alpha included)
        if demod is None:
            gw = geff
        elif ctx.own_demod:
            from ..grad_allreduce import claim_destination
            gw = _weight_demod_bwd(geff, w, demod, ctx.demod_alpha, out=claim_destination(w))
        else:
            gw = geff * demod[:, None, None, None]
            gd = (geff * w).sum(dim=(1, 2, 3))
    return gx, gs, gw, gd


class StyledModConv(Function):
    """StyledConv's plain (stride-1) form as ONE forward kernel (sae_modconv2d_fwd_noise_bias_act_f32):

        out = lrelu((conv(x * s, W * alpha * demod) + noise_weight * noise) + bias, slope) * scale

    i.e. ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU (stylegan2_layers.py:398-405) with the noise map read from LDS in the
    conv's epilogue -- the conv's raw output never exists in HBM.  Backward: the noise + bias + activation backward kernel on the
    saved OUTPUT (mask, bias / noise-strength gradients), then ModulatedConv's backward.  First-order only, like ModulatedConv."""

    @staticmethod
    def forward(ctx, x, s, w, noise, noise_weight, bias, geom, demod_eps, demod_alpha, slope, scale, act_ticket=None,
                input_ticket=None, grad_scale_ticket=None):
        ctx.set_materialize_grads(False)
        lib = hip_lib.get()
        ctx.grad_scale_ticket = grad_scale_ticket
        if grad_scale_ticket is not None:
            grad_scale_ticket.armed = True
        ctx.act_ticket, ctx.input_ticket = act_ticket, (input_ticket if (input_ticket is not None and input_ticket.armed) else None)
        ctx.own_demod = True
        ctx.demod_alpha = demod_alpha
        demod = _weight_demod(w, demod_alpha, demod_eps) if demod_eps is not None else None
        x, w, s, noise = x.contiguous(), w.contiguous(), s.contiguous(), noise.contiguous()
        if act_ticket is not None:
            act_ticket.arm(noise, slope, scale)      # the CONTIGUOUS map: the consumer's backward hands it to a kernel
        lib.check(x, w, s, noise, noise_weight, bias, demod)
        if winograd.eligible(geom, winograd.FWD):
            ctx.factor_tag = ("demod", float(demod_alpha), float(demod_eps)) if demod_eps is not None else ()
            out = winograd.conv(x, w, geom, x_scale=s, row_scale=demod, noise=noise, noise_weight=noise_weight, bias=bias,
                                act=(slope, scale), factor_tag=ctx.factor_tag)
            ctx.save_for_backward(x, s, w, demod, out, noise)
            ctx.cfg = (geom, slope, scale, bias is not None)
            return out
        d = geom.desc()
        mod = hip_lib.ConvMod(s.data_ptr(), None, hip_lib.ptr(demod), None)
        n_ws = lib.query("conv2d_workspace", C.byref(d), SAE_CONV_FWD)
        ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=x.device)
        out = torch.empty((geom.n, geom.m, geom.oh, geom.ow), dtype=torch.float32, device=x.device)
        ctx.factor_tag = ("demod", float(demod_alpha), float(demod_eps)) if demod_eps is not None else ()
        prepared = weight_prep.attach(lib, d, mod, SAE_CONV_FWD, w, geom.alpha, ctx.factor_tag, gkey=geom.key)   # noqa: F841
        lib.call("modconv2d_fwd_noise_bias_act_f32", x.data_ptr(), w.data_ptr(), noise.data_ptr(), noise_weight.data_ptr(),
                 hip_lib.ptr(bias), out.data_ptr(), C.byref(d), C.byref(mod), geom.alpha, float(slope), float(scale),
                 ws.data_ptr(), n_ws, lib.stream(x))
        ctx.save_for_backward(x, s, w, demod, out, noise)
        ctx.cfg = (geom, slope, scale, bias is not None)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, s, w, demod, out, noise = ctx.saved_tensors
        geom, slope, scale, has_bias = ctx.cfg
        need_w = ctx.needs_input_grad[2] and _Flags.weight_grads
        if gout is None:
            gnw = torch.zeros(1, dtype=out.dtype, device=out.device) if ctx.needs_input_grad[4] else None
            gb = torch.zeros(out.shape[1], dtype=out.dtype, device=out.device) if (has_bias and ctx.needs_input_grad[5]) else None
            return None, None, (torch.zeros_like(w) if need_w else None), None, gnw, gb, None, None, None, None, None, None, None, None
        from .modulate import NoiseBiasActBackward
        taken = ctx.act_ticket.take(gout) if ctx.act_ticket is not None else None
        if taken is None:
            # a merge that is this activation's only consumer may have left its 1/sqrt(2) for this kernel (GradScaleTicket)
            fold = ctx.grad_scale_ticket.take(gout) if ctx.grad_scale_ticket is not None else 1.0
            taken = NoiseBiasActBackward.apply(gout, out, noise, slope, scale * fold)
        g_pre, gb, gnw = taken
        gx, gs, gw, _ = _modconv_backward(ctx, g_pre, x, s, w, demod, geom, False, ctx.input_ticket)
        return gx, gs, gw, None, gnw, (gb if has_bias else None), None, None, None, None, None, None, None, None


def styled_modulated_conv2d(input, style_scale, weight, noise, noise_weight, bias, padding=0, alpha=1.0, demod_eps=None,
                            out_scale=1.0, negative_slope=0.2, scale=2 ** 0.5, act_ticket=None, input_ticket=None,
                            grad_scale_ticket=None):
    """StyledConv's plain form (ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU) as one node; arguments as
    modulated_conv2d plus noise [N, 1, H, W], noise_weight [1], bias [O] or None."""
    _check_weight(weight)
    n, c_in, h, w = input.shape
    o, i2, k, _ = weight.shape
    if i2 != c_in or tuple(style_scale.shape) != (n, c_in):
        raise hip_lib.SaeError("styled_modulated_conv2d: input %s, style %s, weight %s do not fit" % (
            tuple(input.shape), tuple(style_scale.shape), tuple(weight.shape)))
    geom = _Geom(n, c_in, h, w, o, k, 1, padding, False, alpha)
    if tuple(noise.shape) != (n, 1, geom.oh, geom.ow):
        raise hip_lib.SaeError("noise must be [N, 1, H, W] = %s, got %s" % ((n, 1, geom.oh, geom.ow), tuple(noise.shape)))
    geom.alpha = float(alpha * out_scale)
    return StyledModConv.apply(input, style_scale, weight, noise, noise_weight, bias, geom, demod_eps, float(alpha),
                               negative_slope, scale, act_ticket, input_ticket, grad_scale_ticket)


def modulated_conv2d(input, style_scale, weight, demod=None, padding=0, alpha=1.0, transposed=False, demod_eps=None,
                     out_scale=1.0):
    """The modulated conv of ModulatedConv2d in its dense-equivalent form.  input [N, I, H, W], style_scale [N, I],
    weight [O, I, k, k], demod [O] or None; transposed=True is the stride-2 upsampling form (output (2H+1) x (2W+1)).
    demod_eps (instead of demod): the demodulation factor rsqrt(sum (alpha w)^2 + demod_eps) is computed, and differentiated,
    inside the node.  out_scale: an extra factor on the output (the conv is linear: a constant pulled out of style_scale)."""
    _check_weight(weight)
    n, c_in, h, w = input.shape
    o, i2, k, _ = weight.shape
    if i2 != c_in or tuple(style_scale.shape) != (n, c_in):
        raise hip_lib.SaeError("modulated_conv2d: input %s, style %s, weight %s do not fit" % (
            tuple(input.shape), tuple(style_scale.shape), tuple(weight.shape)))
    if demod_eps is not None and demod is not None:
        raise hip_lib.SaeError("modulated_conv2d: pass demod or demod_eps, not both")
    if transposed:
        oh, ow = (h - 1) * 2 + k, (w - 1) * 2 + k
        geom = _Geom(n, o, oh, ow, c_in, k, 2, 0, True, alpha)
        assert (geom.oh, geom.ow) == (h, w)
    else:
        geom = _Geom(n, c_in, h, w, o, k, 1, padding, False, alpha)
    # the kernels scale by geom.alpha = alpha * out_scale; the demodulation factor is defined on alpha * w
    geom.alpha = float(alpha * out_scale)
    return ModulatedConv.apply(input, style_scale, weight, demod, geom, transposed, demod_eps, float(alpha))


def _check_weight(weight):
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3]:
        raise hip_lib.SaeError("conv weight must be [M, C, k, k], got %s" % (tuple(weight.shape),))
    if weight.shape[2] not in (1, 3):
        raise hip_lib.SaeError("MI355X conv kernels cover k in {1, 3}; got k=%d" % weight.shape[2])


def conv2d(input, weight, bias=None, stride=1, padding=0, alpha=1.0):
    """F.conv2d semantics for the shapes of the hot path: weight [M, C, k, k], k in {1, 3},
    stride in {1, 2}, symmetric zero ``padding`` < k.  Returns alpha * conv(input, weight) + bias."""
    _check_weight(weight)
    if stride not in (1, 2):
        raise hip_lib.SaeError("MI355X conv kernels cover stride in {1, 2}; got %r" % (stride,))
    n, c, h, w = input.shape
    m, c2, k, _ = weight.shape
    if c2 != c:
        raise hip_lib.SaeError("conv2d: input has %d channels, weight expects %d" % (c, c2))
    geom = _Geom(n, c, h, w, m, k, stride, padding, False, alpha)
    out = ConvForward.apply(input, weight, geom)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def conv2d_bias_act(input, weight, bias=None, stride=1, padding=0, alpha=1.0, negative_slope=0.2, scale=2 ** 0.5):
    """fused_leaky_relu(conv2d(input, weight) * alpha, bias, negative_slope, scale) as one kernel."""
    _check_weight(weight)
    if stride not in (1, 2):
        raise hip_lib.SaeError("MI355X conv kernels cover stride in {1, 2}; got %r" % (stride,))
    n, c, h, w = input.shape
    m, c2, k, _ = weight.shape
    if c2 != c:
        raise hip_lib.SaeError("conv2d: input has %d channels, weight expects %d" % (c, c2))
    geom = _Geom(n, c, h, w, m, k, stride, padding, False, alpha)
    return ConvBiasAct.apply(input, weight, bias, geom, negative_slope, scale)


def conv_transpose2d(input, weight, stride=2, alpha=1.0):
    """Stride-2, padding-0 transposed convolution (stylegan2_layers.py:306) with the weight kept in
    the parameter's own [M_out, C_in, k, k] orientation:
        out[n, m, 2*i + ky, 2*j + kx] += alpha * weight[m, c, ky, kx] * input[n, c, i, j]
    i.e. the data gradient of a stride-2 conv that maps M_out -> C_in channels."""
    _check_weight(weight)
    if stride != 2:
        raise hip_lib.SaeError("conv_transpose2d: only stride 2 is on the hot path")
    n, c_in, h, w = input.shape
    m_out, c2, k, _ = weight.shape
    if c2 != c_in:
        raise hip_lib.SaeError("conv_transpose2d: input has %d channels, weight expects %d" % (c_in, c2))
    oh, ow = (h - 1) * 2 + k, (w - 1) * 2 + k
    # forward-orientation problem: x side = the large output (m_out channels), y side = input
    geom = _Geom(n, m_out, oh, ow, c_in, k, 2, 0, True, alpha)
    assert (geom.oh, geom.ow) == (h, w)
    return ConvDataGrad.apply(input, weight, geom)


# ------------------------------------------------------------------------------------------------
# linear
# ------------------------------------------------------------------------------------------------
def _gemm(a, b, bias, ta, tb, alpha):
    """alpha * op(a) @ op(b) (+ bias) for 2-D contiguous a, b; op = transpose when the flag is set."""
    lib = hip_lib.get()
    a = a.contiguous()
    b = b.contiguous()
    bias = bias.contiguous() if bias is not None else None
    lib.check(a, b, bias)
    m, k = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    k2, n = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
    if k != k2:
        raise hip_lib.SaeError("gemm: inner dimensions differ (%d vs %d)" % (k, k2))
    a_si, a_sk = (1, a.shape[1]) if ta else (a.shape[1], 1)
    b_sk, b_sj = (1, b.shape[1]) if tb else (b.shape[1], 1)
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    # skinny products (16 - 128 rows against 2048 x 2048 ... 8192 x 512 weights): the contraction split across workgroups
    n_ws = lib.query("gemm_workspace", m, n, k)
    ws = torch.empty(n_ws, dtype=torch.float32, device=a.device) if n_ws > 0 else None
    lib.call("gemm_ws_f32", a.data_ptr(), b.data_ptr(), hip_lib.ptr(bias), out.data_ptr(), m, n, k, a_si, a_sk, b_sk,
             b_sj, n, float(alpha), hip_lib.ptr(ws), n_ws, lib.stream(a))
    return out


class MatMul(Function):
    """alpha * op(a) @ op(b) + bias, closed under differentiation (each gradient is a MatMul)."""

    @staticmethod
    def forward(ctx, a, b, bias, ta, tb, alpha):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        ctx.cfg = (ta, tb, alpha)
        ctx.save_for_backward(a, b)
        return _gemm(a, b, bias, ta, tb, alpha)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        ta, tb, alpha = ctx.cfg
        if gc is None:
            gbias = None
            if ctx.needs_input_grad[2]:
                gbias = torch.zeros(b.shape[0] if tb else b.shape[1], dtype=b.dtype, device=b.device)
            return None, _zero_param_grad(ctx, 1, b), gbias, None, None, None
        ga = gb = gbias = None
        if ctx.needs_input_grad[0]:
            if not ta:
                ga = MatMul.apply(gc, b, None, False, not tb, alpha)     # gc @ op(b)^T
            else:
                ga = MatMul.apply(b, gc, None, tb, True, alpha)          # op(b) @ gc^T
        if ctx.needs_input_grad[1] and _Flags.weight_grads:
            if not tb:
                gb = MatMul.apply(a, gc, None, not ta, False, alpha)     # op(a)^T @ gc
            else:
                gb = MatMul.apply(gc, a, None, True, ta, alpha)          # gc^T @ op(a)
        if ctx.needs_input_grad[2]:
            gbias = gc.sum(0)
        return ga, gb, gbias, None, None, None


def linear(input, weight, bias=None, alpha=1.0):
    """F.linear semantics: alpha * input @ weight^T + bias, input [..., in], weight [out, in]."""
    lead = input.shape[:-1]
    x2 = input.reshape(-1, input.shape[-1])
    out = MatMul.apply(x2, weight, bias, False, True, alpha)
    return out.view(*lead, weight.shape[0])
