"""``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`` on the gfx950 K1 kernel.

Mirrors the operator interface of the reference (models/networks/stylegan2_op/upfirdn2d.py:150-159:
same name, argument meaning, NCHW in / NCHW out, same pad on x and y) and its differentiation
contract: twice differentiable w.r.t. ``input`` (R1 regularisation needs the double backward,
swapping_autoencoder_model.py:143-148), no gradient for ``kernel``.

The adjoint of an (up, down, pad) call with taps k is an upfirdn2d call with flipped taps,
(up, down) exchanged and the pads of upfirdn2d.py:116-121; its adjoint is the forward call again,
so two Function classes close the chain at any order.
"""
import weakref

import torch
from torch.autograd import Function

from .. import hip_lib


def _out_size(size, up, down, pad0, pad1, taps):
    return (size * up + pad0 + pad1 - taps) // down + 1


def _run(x, taps, up, down, pad):
    """x: [N, C, H, W] -> [N, C, H', W'] through sae_upfirdn2d_f32 (major = N*C, minor = 1)."""
    lib = hip_lib.get()
    x = x.contiguous()
    taps = taps.contiguous()
    lib.check(x, taps)
    n, c, h, w = x.shape
    kh, kw = taps.shape
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    oh = _out_size(h, up_y, down_y, py0, py1, kh)
    ow = _out_size(w, up_x, down_x, px0, px1, kw)
    if oh < 1 or ow < 1:
        raise hip_lib.SaeError("upfirdn2d: empty output %dx%d" % (oh, ow))
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device)
    lib.call("upfirdn2d_f32", x.data_ptr(), taps.data_ptr(), y.data_ptr(), n * c, h, w, 1, kh, kw,
             up_x, up_y, down_x, down_y, px0, px1, py0, py1, lib.stream(x))
    return y


def _adjoint_pad(in_hw, out_hw, taps_hw, up, down, pad):
    """Pads of the adjoint call (reference upfirdn2d.py:116-121)."""
    (ih, iw), (oh, ow), (kh, kw) = in_hw, out_hw, taps_hw
    up_x, up_y = up
    down_x, down_y = down
    px0, _, py0, _ = pad
    gx0 = kw - px0 - 1
    gy0 = kh - py0 - 1
    gx1 = iw * up_x - ow * down_x + px0 - up_x + 1
    gy1 = ih * up_y - oh * down_y + py0 - up_y + 1
    return (gx0, gx1, gy0, gy1)


_FLIPPED = {}


def _flipped(kernel):
    """The taps flipped in both directions (upfirdn2d.py:119 of the reference flips them in every backward call: 52
    `aten::flip` launches per iteration).  The FIR taps are module buffers that never change: one flip per buffer object,
    re-done if the buffer is modified in place; entries die with their tensor (weak reference, identity checked, so a
    recycled address or id cannot alias another tensor's taps)."""
    entry = _FLIPPED.get(id(kernel))
    if entry is not None and entry[0]() is kernel and entry[1] == kernel._version:
        return entry[2]
    if len(_FLIPPED) > 64:
        for k in [k for k, e in _FLIPPED.items() if e[0]() is None]:
            del _FLIPPED[k]
        if len(_FLIPPED) > 64:
            _FLIPPED.clear()
    out = torch.flip(kernel, [0, 1])
    _FLIPPED[id(kernel)] = (weakref.ref(kernel), kernel._version, out)
    return out


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        out = _run(input, kernel, up, down, pad)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, tuple(input.shape[2:]), tuple(out.shape[2:]))
        return out

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output is None:
            return None, None, None, None, None
        kernel, = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = UpFirDn2dBackward.apply(grad_output, kernel, *ctx.cfg)
        return grad_input, None, None, None, None


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, up, down, pad, in_hw, out_hw):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        g_pad = _adjoint_pad(in_hw, out_hw, tuple(kernel.shape), up, down, pad)
        flipped = _flipped(kernel)
        grad_input = _run(grad_output, flipped, down, up, g_pad)   # (up, down) exchanged
        assert tuple(grad_input.shape[2:]) == tuple(in_hw), (grad_input.shape, in_hw)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad)
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        if gradgrad_input is None:
            return None, None, None, None, None, None, None
        kernel, = ctx.saved_tensors
        up, down, pad = ctx.cfg
        gradgrad_out = None
        if ctx.needs_input_grad[0]:
            gradgrad_out = UpFirDn2d.apply(gradgrad_input, kernel, up, down, pad)
        return gradgrad_out, None, None, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


class BlurNoiseBiasAct(Function):
    """upfirdn2d(x, kernel, pad) -> NoiseInjection -> FusedLeakyReLU as ONE kernel (sae_upfirdn2d_noise_bias_act_f32): the tail of
    StyledConv's upsampling form (stylegan2_layers.py:313-321, :398-405).  Backward: the noise + bias + activation backward on the
    saved OUTPUT, then the blur's adjoint.  First-order only (like the modulated conv in front of it)."""

    @staticmethod
    def forward(ctx, input, kernel, pad, noise, noise_weight, bias, negative_slope, scale, act_ticket=None):
        ctx.set_materialize_grads(False)
        lib = hip_lib.get()
        ctx.act_ticket = act_ticket
        input, kernel, noise = input.contiguous(), kernel.contiguous(), noise.contiguous()
        if act_ticket is not None:
            act_ticket.arm(noise, negative_slope, scale)      # the CONTIGUOUS map: the consumer's backward hands it to a kernel
        lib.check(input, kernel, noise, noise_weight, bias)
        n, c, h, w = input.shape
        kh, kw = kernel.shape
        px0, px1, py0, py1 = pad
        oh, ow = _out_size(h, 1, 1, py0, py1, kh), _out_size(w, 1, 1, px0, px1, kw)
        if tuple(noise.shape) != (n, 1, oh, ow):
            raise hip_lib.SaeError("noise must be [N, 1, H, W] = %s, got %s" % ((n, 1, oh, ow), tuple(noise.shape)))
        out = torch.empty((n, c, oh, ow), dtype=input.dtype, device=input.device)
        lib.call("upfirdn2d_noise_bias_act_f32", input.data_ptr(), kernel.data_ptr(), out.data_ptr(), n * c, h, w, kh, kw,
                 px0, px1, py0, py1, noise.data_ptr(), noise_weight.data_ptr(), hip_lib.ptr(bias), c, float(negative_slope),
                 float(scale), lib.stream(input))
        ctx.save_for_backward(kernel, out, noise)
        ctx.cfg = (pad, (h, w), (oh, ow), negative_slope, scale, bias is not None)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        kernel, out, noise = ctx.saved_tensors
        pad, in_hw, out_hw, negative_slope, scale, has_bias = ctx.cfg
        if grad_output is None:
            gw = torch.zeros(1, dtype=out.dtype, device=out.device) if ctx.needs_input_grad[4] else None
            gb = torch.zeros(out.shape[1], dtype=out.dtype, device=out.device) if (has_bias and ctx.needs_input_grad[5]) else None
            return None, None, None, None, gw, gb, None, None, None
        from .modulate import NoiseBiasActBackward
        taken = ctx.act_ticket.take(grad_output) if ctx.act_ticket is not None else None
        g_pre, gb, gw = taken if taken is not None else NoiseBiasActBackward.apply(grad_output, out, noise, negative_slope, scale)
        gx = UpFirDn2dBackward.apply(g_pre, kernel, (1, 1), (1, 1), pad, in_hw, out_hw) if ctx.needs_input_grad[0] else None
        return gx, None, None, None, gw, (gb if has_bias else None), None, None, None


def blur_noise_bias_act(input, kernel, pad, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5, act_ticket=None):
    """leaky_relu((upfirdn2d(input, kernel, pad=pad) + noise_weight * noise) + bias) * scale in one kernel; taps <= 4 x 4."""
    return BlurNoiseBiasAct.apply(input, kernel, (pad[0], pad[1], pad[0], pad[1]), noise, noise_weight, bias, negative_slope, scale,
                                  act_ticket)
