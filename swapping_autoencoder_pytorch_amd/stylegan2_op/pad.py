"""Reflection padding on the gfx950 kernels (csrc/pad.hip): nn.ReflectionPad2d / F.pad(mode="reflect") as used by
the encoder's ConvLayer and Blur (reference: models/networks/stylegan2_layers.py:57-63, :100-105, :643).  Forward
and adjoint are each other's backward (linear op), so the Function pair is differentiable to any order, and both
are deterministic (ATen's reflection_pad2d_backward accumulates with atomics)."""
import torch
from torch import nn
from torch.autograd import Function

from .. import hip_lib


def _pads(pad):
    if isinstance(pad, int):
        return (pad, pad, pad, pad)
    pad = tuple(int(p) for p in pad)
    if len(pad) != 4:
        raise hip_lib.SaeError("reflection pad wants (left, right, top, bottom), got %r" % (pad,))
    return pad


def _call(name, src, out_hw, in_hw, pads):
    lib = hip_lib.get()
    src = src.contiguous()
    lib.check(src)
    out = torch.empty(src.shape[:-2] + out_hw, dtype=src.dtype, device=src.device)
    planes = 1
    for s in src.shape[:-2]:
        planes *= s
    lib.call(name, src.data_ptr(), out.data_ptr(), planes, in_hw[0], in_hw[1], pads[0], pads[1], pads[2], pads[3],
             lib.stream(src))
    return out


class ReflectPadFunction(Function):
    @staticmethod
    def forward(ctx, x, pads):
        ctx.set_materialize_grads(False)
        h, w = x.shape[-2:]
        ctx.cfg = (pads, (h, w))
        return _call("reflect_pad_f32", x, (h + pads[2] + pads[3], w + pads[0] + pads[1]), (h, w), pads)

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None
        pads, hw = ctx.cfg
        return ReflectPadAdjoint.apply(gy, pads, hw), None


class ReflectPadAdjoint(Function):
    @staticmethod
    def forward(ctx, gy, pads, hw):
        ctx.set_materialize_grads(False)
        ctx.cfg = pads
        return _call("reflect_pad_adj_f32", gy, hw, hw, pads)

    @staticmethod
    def backward(ctx, ggx):
        if ggx is None:
            return None, None, None
        return ReflectPadFunction.apply(ggx, ctx.cfg), None, None


def reflect_pad(x, pad):
    """F.pad(x, (left, right, top, bottom), mode="reflect") for [..., H, W] fp32 tensors."""
    return ReflectPadFunction.apply(x, _pads(pad))


class ReflectionPad2d(nn.ReflectionPad2d):
    """Drop-in for nn.ReflectionPad2d (same constructor, no parameters or buffers)."""

    def forward(self, input):
        lib = hip_lib.get()
        if input.dtype == torch.float32 and input.dim() >= 3 and (input.is_cuda or not lib.device_only):
            return reflect_pad(input, self.padding)
        return super().forward(input)
