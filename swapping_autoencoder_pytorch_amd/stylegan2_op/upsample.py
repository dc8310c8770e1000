"""Bilinear x2 upsampling fused with the residual add and scale of the generator's upsampling
blocks: ``alpha * (F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) + res)``
(reference: models/networks/generator.py:51-53) on one gfx950 kernel instead of three elementwise
passes over the full-resolution tensor.  The operator is linear, so forward and adjoint close the
differentiation chain at any order."""
import torch
from torch.autograd import Function

from .. import hip_lib


def _up(x, res, alpha):
    lib = hip_lib.get()
    x = x.contiguous()
    res = res.contiguous() if res is not None else None
    lib.check(x, res)
    n, c, h, w = x.shape
    if res is not None and tuple(res.shape) != (n, c, 2 * h, 2 * w):
        raise hip_lib.SaeError("upsample2x_add: residual %s does not match 2x of %s" % (tuple(res.shape), tuple(x.shape)))
    y = torch.empty((n, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    lib.call("upsample2x_bilinear_add_f32", x.data_ptr(), hip_lib.ptr(res), y.data_ptr(), n * c, h, w, float(alpha),
             lib.stream(x))
    return y


def _down(gy, alpha):
    lib = hip_lib.get()
    gy = gy.contiguous()
    lib.check(gy)
    n, c, oh, ow = gy.shape
    gx = torch.empty((n, c, oh // 2, ow // 2), dtype=gy.dtype, device=gy.device)
    lib.call("upsample2x_bilinear_bwd_f32", gy.data_ptr(), gx.data_ptr(), n * c, oh // 2, ow // 2, float(alpha),
             lib.stream(gy))
    return gx


class Upsample2xAdd(Function):
    @staticmethod
    def forward(ctx, x, res, alpha, res_ticket=None):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        ctx.alpha = alpha
        ctx.res_ticket = res_ticket if (res_ticket is not None and res_ticket.armed) else None
        return _up(x, res, alpha)

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None, None
        gx = Upsample2xAdjoint.apply(gy, ctx.alpha) if ctx.needs_input_grad[0] else None
        gres = None
        if ctx.needs_input_grad[1]:
            # the producer of `res` applies alpha inside its own backward kernel when the block handed over a ticket
            # (stylegan2_op.modulate.GradScaleTicket); otherwise one pass over the full-resolution gradient
            gres = ctx.res_ticket.offer(gy.contiguous(), ctx.alpha) if (ctx.res_ticket is not None and not torch.is_grad_enabled()) \
                else gy * ctx.alpha
        return gx, gres, None, None


class Upsample2xAdjoint(Function):
    @staticmethod
    def forward(ctx, gy, alpha):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        ctx.alpha = alpha
        return _down(gy, alpha)

    @staticmethod
    def backward(ctx, ggx):
        if ggx is None:
            return None, None
        return (Upsample2xAdd.apply(ggx, None, ctx.alpha) if ctx.needs_input_grad[0] else None), None


def upsample2x_add(x, res=None, alpha=1.0, res_ticket=None):
    return Upsample2xAdd.apply(x, res, alpha, res_ticket)


class AddScale(Function):
    """alpha * (a + b) in one pass: the residual merge ``(out + skip) / sqrt(2)`` of ResBlock
    (stylegan2_layers.py:689) and of the generator's blocks (generator.py:36)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        lib = hip_lib.get()
        a = a.contiguous()
        b = b.contiguous()
        lib.check(a, b)
        if a.shape != b.shape:
            raise hip_lib.SaeError("add_scale: shapes differ %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        y = torch.empty_like(a)
        lib.call("add_scale_f32", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), float(alpha), lib.stream(a))
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None
        g = gy * ctx.alpha          # one pass; both inputs receive the same tensor
        return (g if ctx.needs_input_grad[0] else None), (g if ctx.needs_input_grad[1] else None), None


def add_scale(a, b, alpha):
    return AddScale.apply(a, b, alpha)
