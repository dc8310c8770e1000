"""Fused elementwise glue of the generator's StyledConv (csrc/modulate.hip).

``noise_bias_act``  = NoiseInjection.forward followed by FusedLeakyReLU.forward
                      (models/networks/stylegan2_layers.py:340-351 and :54-65) in one pass; its backward
                      produces grad_input, grad_bias and the gradient of the scalar noise weight in one pass.
``plane_scale``     = the style modulation ``x * s[:, :, None, None]`` of ModulatedConv2d
                      (stylegan2_layers.py:280-286); the forward is a plain broadcast multiply, the backward
                      (grad_x = g * s, grad_s = sum_hw g * x) is one fused pass.
Both stay twice differentiable (the second-order graphs are expressed with the first-order ops).
"""
import torch
from torch.autograd import Function

from .. import hip_lib
from .fused_act import _bias_act


def fusable(x):
    """The fused kernels want [N, C, H, W] planes of a multiple of 4 elements."""
    return x.dim() == 4 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[2] * x.shape[3] >= 4


class ActTicket:
    """Hand-over between two nodes of ONE generator block (networks/generator.py): conv1's node (the producer: its output is a
    noise + bias + leaky-ReLU activation) and conv2's node (the ONLY consumer of that output).  The consumer's backward runs
    `plane_scale_dot` and the producer's activation backward as one kernel (sae_plane_scale_dot_act_f32) and leaves the bias /
    noise-strength gradients here; the producer's backward recognises the pre-processed gradient by its address and skips its
    own activation pass.  Only block code that knows the activation has a single consumer may pass a ticket.  Side effect of the
    address-based hand-over: a tensor hook or `retain_grad()` on the handed-over activation observes the PRE-activation gradient
    (what the consumer's fused kernel wrote), not d(loss)/d(activation)."""
    __slots__ = ("armed", "noise", "slope", "scale", "gb", "gnw", "grad_ptr", "done")

    def __init__(self):
        self.armed = self.done = False
        self.noise = self.gb = self.gnw = None
        self.slope = self.scale = self.grad_ptr = None

    def arm(self, noise, slope, scale):
        self.armed, self.noise, self.slope, self.scale = True, noise, float(slope), float(scale)

    def take(self, grad_output):
        """(g_pre, gb, gnw) when `grad_output` is the consumer's pre-processed gradient, else None."""
        if not self.done:
            return None
        self.done = False
        if grad_output.data_ptr() != self.grad_ptr:
            raise hip_lib.SaeError("ActTicket: the activation has a second consumer (its gradient was re-summed); "
                                   "run with SAE_STYLED_FUSED=0")
        out = (grad_output, self.gb, self.gnw)
        self.gb = self.gnw = None
        return out


class GradScaleTicket:
    """The other direction inside an upsampling block: the merge `alpha * (up2x(skip) + res)` is the ONLY consumer of conv2's
    activated output `res`.  Its backward hands the incoming gradient on untouched and leaves `alpha` here; conv2's node folds it
    into the scale of its activation backward -- the `gy * alpha` pass over the full-resolution tensor disappears."""
    __slots__ = ("armed", "alpha", "grad_ptr")

    def __init__(self):
        self.armed, self.alpha, self.grad_ptr = False, None, None

    def offer(self, gy, alpha):
        self.alpha, self.grad_ptr = float(alpha), gy.data_ptr()
        return gy

    def take(self, grad_output):
        """the factor the producer must apply to `grad_output` (1.0 when the merge did not hand over)"""
        if self.alpha is None:
            return 1.0
        alpha, ptr = self.alpha, self.grad_ptr
        self.alpha = self.grad_ptr = None
        if grad_output.data_ptr() != ptr:
            raise hip_lib.SaeError("GradScaleTicket: the activation has a second consumer (its gradient was re-summed); "
                                   "run with SAE_STYLED_FUSED=0")
        return alpha


def plane_scale_dot_act(g, x, s, ticket):
    """plane_scale_backward(g, x, s) fused with the producer's noise + bias + activation backward: returns (g_pre, gs) and
    leaves gb / gnw in the ticket."""
    lib = hip_lib.get()
    g = g.contiguous()
    lib.check(g, x, s, ticket.noise)
    n, c, h, w = x.shape
    gx = torch.empty_like(g)
    gs = torch.empty((n, c), dtype=x.dtype, device=x.device)
    gb = torch.empty(c, dtype=x.dtype, device=x.device)
    gw = torch.empty(1, dtype=x.dtype, device=x.device)
    n_ws = lib.query("plane_scale_dot_act_workspace", n, c)
    ws = torch.empty(max(n_ws, 1), dtype=x.dtype, device=x.device)
    lib.call("plane_scale_dot_act_f32", g.data_ptr(), x.data_ptr(), s.data_ptr(), hip_lib.ptr(ticket.noise), gx.data_ptr(),
             gs.data_ptr(), gb.data_ptr(), gw.data_ptr(), ws.data_ptr(), n_ws, n, c, h * w, ticket.slope, ticket.scale,
             lib.stream(x))
    ticket.gb, ticket.gnw, ticket.grad_ptr, ticket.done = gb, gw, gx.data_ptr(), True
    return gx, gs


class NoiseBiasActFunction(Function):
    @staticmethod
    def forward(ctx, input, noise, noise_weight, bias, negative_slope, scale):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        lib = hip_lib.get()
        input = input.contiguous()
        noise = noise.contiguous()
        lib.check(input, noise, noise_weight, bias)
        n, c, h, w = input.shape
        if noise.shape != (n, 1, h, w):
            raise hip_lib.SaeError("noise must be [N, 1, H, W] = %s, got %s" % ((n, 1, h, w), tuple(noise.shape)))
        out = torch.empty_like(input)
        lib.call("noise_bias_act_f32", input.data_ptr(), noise.data_ptr(), noise_weight.data_ptr(), hip_lib.ptr(bias),
                 out.data_ptr(), n, c, h * w, float(negative_slope), float(scale), lib.stream(input))
        ctx.save_for_backward(out, noise)
        ctx.cfg = (negative_slope, scale, bias is not None)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output is None:      # undefined = zero: explicit zeros for the two parameters only
            out, noise = ctx.saved_tensors
            gw = torch.zeros(1, dtype=out.dtype, device=out.device) if ctx.needs_input_grad[2] else None
            gb = torch.zeros(out.shape[1], dtype=out.dtype, device=out.device) if (ctx.cfg[2] and ctx.needs_input_grad[3]) else None
            return None, None, gw, gb, None, None
        out, noise = ctx.saved_tensors
        negative_slope, scale, has_bias = ctx.cfg
        gx, gb, gw = NoiseBiasActBackward.apply(grad_output, out, noise, negative_slope, scale)
        return gx, None, gw, (gb if has_bias else None), None, None


class NoiseBiasActBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, noise, negative_slope, scale):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        lib = hip_lib.get()
        grad_output = grad_output.contiguous()
        lib.check(grad_output, out, noise)
        n, c, h, w = out.shape
        gx = torch.empty_like(grad_output)
        gb = torch.empty(c, dtype=out.dtype, device=out.device)
        gw = torch.empty(1, dtype=out.dtype, device=out.device)
        n_ws = lib.query("noise_bias_act_bwd_workspace", n, c, h * w)
        ws = torch.empty(max(n_ws, 1), dtype=out.dtype, device=out.device)
        lib.call("noise_bias_act_bwd_f32", grad_output.data_ptr(), out.data_ptr(), noise.data_ptr(), gx.data_ptr(),
                 gb.data_ptr(), gw.data_ptr(), ws.data_ptr(), n_ws, n, c, h * w, float(negative_slope), float(scale),
                 lib.stream(out))
        ctx.save_for_backward(out, noise)
        ctx.cfg = (negative_slope, scale)
        return gx, gb, gw

    @staticmethod
    def backward(ctx, gg_x, gg_b, gg_w):
        if gg_x is None and gg_b is None and gg_w is None:
            return None, None, None, None, None
        # gx, gb and gw are linear in grad_output through the same mask:
        # d/d(grad_output) = (out > 0 ? 1 : slope) * scale * (gg_x + gg_b[c] + gg_w * noise)
        out, noise = ctx.saved_tensors
        negative_slope, scale = ctx.cfg
        t = gg_x if gg_x is not None else torch.zeros_like(out)
        if gg_w is not None:
            t = t + gg_w * noise
        return _bias_act(t, gg_b, out, 1, negative_slope, scale), None, None, None, None


def noise_bias_act(input, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    """leaky_relu((input + noise_weight * noise) + bias) * scale; noise: [N, 1, H, W], noise_weight: 1 element."""
    return NoiseBiasActFunction.apply(input, noise, noise_weight, bias, negative_slope, scale)


class PlaneScaleFunction(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        ctx.save_for_backward(x, s)
        return x * s[:, :, None, None]

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output is None:
            return None, None
        x, s = ctx.saved_tensors
        return PlaneScaleBackward.apply(grad_output, x, s)


class PlaneScaleBackward(Function):
    @staticmethod
    def forward(ctx, g, x, s):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        lib = hip_lib.get()
        g = g.contiguous()
        x = x.contiguous()
        s = s.contiguous()
        lib.check(g, x, s)
        n, c, h, w = x.shape
        gx = torch.empty_like(g)
        gs = torch.empty_like(s)
        lib.call("plane_scale_dot_f32", g.data_ptr(), x.data_ptr(), s.data_ptr(), gx.data_ptr(), gs.data_ptr(), n * c,
                 h * w, lib.stream(x))
        ctx.save_for_backward(g, x, s)
        return gx, gs

    @staticmethod
    def backward(ctx, gg_x, gg_s):
        if gg_x is None and gg_s is None:
            return None, None, None
        # gx = g * s, gs = <g, x>: bilinear, differentiated with plain tensor ops
        g, x, s = ctx.saved_tensors
        grad_g = grad_x = grad_s = None
        if gg_x is not None:
            grad_g = gg_x * s[:, :, None, None]
            grad_s = (gg_x * g).sum(dim=(2, 3))
        if gg_s is not None:
            t = gg_s[:, :, None, None]
            grad_g = t * x if grad_g is None else grad_g + t * x
            grad_x = t * g
        return grad_g, grad_x, grad_s


def plane_scale_backward(g, x, s):
    """(g * s[:, :, None, None], sum_hw g * x) in one pass (plain tensors, no autograd graph): the backward of the style
    modulation, used by conv2d_gemm.ModulatedConv."""
    lib = hip_lib.get()
    g, x, s = g.contiguous(), x.contiguous(), s.contiguous()
    lib.check(g, x, s)
    n, c, h, w = x.shape
    gx = torch.empty_like(g)
    gs = torch.empty_like(s)
    lib.call("plane_scale_dot_f32", g.data_ptr(), x.data_ptr(), s.data_ptr(), gx.data_ptr(), gs.data_ptr(), n * c, h * w,
             lib.stream(x))
    return gx, gs


def plane_scale(x, s):
    """x * s[:, :, None, None] for x: [N, C, H, W], s: [N, C]."""
    return PlaneScaleFunction.apply(x, s)
