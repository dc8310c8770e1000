"""One-launch forms of the small ATen chains of the train step (csrc/glue.hip): ``util.normalize`` (util/util.py:18-22),
``GeneratorModulation`` (models/networks/generator.py:62-67) and ``gan_loss`` (models/networks/loss.py:10-16).
First-order differentiable: none of them sits on a path that is differentiated twice (the R1 penalties differentiate
D / Dpatch with respect to their INPUT images and use ``pred.sum()``, not ``gan_loss``)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import hip_lib


class L2Normalize(Function):
    @staticmethod
    def forward(ctx, x, eps):
        lib = hip_lib.get()
        x = x.contiguous()
        lib.check(x)
        outer, channels = x.shape[0], x.shape[1]
        inner = x[0, 0].numel() if x.dim() > 2 else 1
        y = torch.empty_like(x)
        lib.call("l2_normalize_f32", x.data_ptr(), y.data_ptr(), outer, channels, inner, float(eps), lib.stream(x))
        ctx.save_for_backward(x)
        ctx.cfg = (outer, channels, inner, float(eps))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        outer, channels, inner, eps = ctx.cfg
        lib = hip_lib.get()
        gy = gy.contiguous()
        lib.check(gy)
        gx = torch.empty_like(x)
        lib.call("l2_normalize_bwd_f32", gy.data_ptr(), x.data_ptr(), gx.data_ptr(), outer, channels, inner, eps, lib.stream(x))
        return gx, None


def l2_normalize(x, eps=1e-8):
    """x * rsqrt(sum(x ** 2, dim=1, keepdim=True) + eps) for [N, C] or [N, C, ...] tensors."""
    return L2Normalize.apply(x, eps)


class PlaneAffine(Function):
    @staticmethod
    def forward(ctx, x, a, b):
        lib = hip_lib.get()
        x, a, b = x.contiguous(), a.contiguous(), b.contiguous()
        lib.check(x, a, b)
        n, c = x.shape[:2]
        hw = x[0, 0].numel()
        if tuple(a.shape) != (n, c) or tuple(b.shape) != (n, c):
            raise hip_lib.SaeError("plane_affine: scale / bias must be [N, C] = %s" % ((n, c),))
        y = torch.empty_like(x)
        lib.call("plane_affine_f32", x.data_ptr(), a.data_ptr(), b.data_ptr(), y.data_ptr(), n * c, hw, lib.stream(x))
        ctx.save_for_backward(x, a)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, a = ctx.saved_tensors
        lib = hip_lib.get()
        g = g.contiguous()
        lib.check(g)
        n, c = x.shape[:2]
        gx, ga, gb = torch.empty_like(x), torch.empty_like(a), torch.empty_like(a)
        lib.call("plane_affine_bwd_f32", g.data_ptr(), x.data_ptr(), a.data_ptr(), gx.data_ptr(), ga.data_ptr(), gb.data_ptr(),
                 n * c, x[0, 0].numel(), lib.stream(x))
        return gx, ga, gb


def plane_affine(x, scale, bias):
    """x * scale[:, :, None, None] + bias[:, :, None, None]"""
    return PlaneAffine.apply(x, scale, bias)


class SoftplusMean(Function):
    @staticmethod
    def forward(ctx, x, sign):
        lib = hip_lib.get()
        x = x.contiguous()
        lib.check(x)
        batch = x.shape[0]
        inner = x[0].numel() if batch > 0 else 1
        y = torch.empty(batch, dtype=x.dtype, device=x.device)
        lib.call("softplus_mean_f32", x.data_ptr(), y.data_ptr(), batch, inner, float(sign), lib.stream(x))
        ctx.save_for_backward(x)
        ctx.cfg = (batch, inner, float(sign))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        batch, inner, sign = ctx.cfg
        lib = hip_lib.get()
        gy = gy.contiguous()
        lib.check(gy)
        gx = torch.empty_like(x)
        lib.call("softplus_mean_bwd_f32", gy.data_ptr(), x.data_ptr(), gx.data_ptr(), batch, inner, sign, lib.stream(x))
        return gx, None


def softplus_mean(x, sign):
    """F.softplus(sign * x).view(B, -1).mean(dim=1)"""
    return SoftplusMean.apply(x, sign)
