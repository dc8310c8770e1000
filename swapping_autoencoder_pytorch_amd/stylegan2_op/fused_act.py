"""``fused_leaky_relu`` / ``FusedLeakyReLU`` on the gfx950 K2 kernels.

Mirrors models/networks/stylegan2_op/fused_act.py:77-96 (same names, arguments, parameter
``bias`` of shape (channel,) initialised to zero) and its differentiation contract (twice
differentiable w.r.t. input and bias).  Differences from the reference's CUDA path, both
deliberate: the backward produces grad_input AND grad_bias in one pass over HBM
(sae_bias_act_bwd_f32) instead of a kernel plus a separate ``sum`` (fused_act.py:32-41), and the
bias gradient reduction is deterministic.
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import hip_lib

_LRELU = 3


def _geometry(x):
    """(step_b, size_b): bias index of flat element i is (i // step_b) % size_b (channel = dim 1)."""
    if x.dim() < 2:
        raise hip_lib.SaeError("fused_leaky_relu expects at least 2 dims, got %d" % x.dim())
    step = 1
    for s in x.shape[2:]:
        step *= s
    return step, x.shape[1]


def _bias_act(x, bias, ref, grad, alpha, scale):
    lib = hip_lib.get()
    x = x.contiguous()
    bias = bias.contiguous() if bias is not None else None
    ref = ref.contiguous() if ref is not None else None
    lib.check(x, bias, ref)
    step, size = _geometry(x)
    if bias is not None and bias.numel() != size:
        raise hip_lib.SaeError("bias has %d elements, channel dim has %d" % (bias.numel(), size))
    y = torch.empty_like(x)
    lib.call("bias_act_f32", x.data_ptr(), hip_lib.ptr(bias), hip_lib.ptr(ref), y.data_ptr(), x.numel(), step, size,
             _LRELU, grad, float(alpha), float(scale), lib.stream(x))
    return y


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        out = _bias_act(input, bias, None, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale, bias is not None)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output is None:      # undefined = zero: nothing for the activation, an explicit zero for the bias
            out, = ctx.saved_tensors
            gb = torch.zeros(out.shape[1], dtype=out.dtype, device=out.device) if (ctx.cfg[2] and ctx.needs_input_grad[1]) else None
            return None, gb, None, None
        out, = ctx.saved_tensors
        negative_slope, scale, has_bias = ctx.cfg
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, negative_slope, scale)
        return grad_input, (grad_bias if has_bias else None), None, None


def bias_act_bwd_raw(grad_output, out, negative_slope, scale):
    """(grad_input, grad_bias) of the leaky-ReLU form in one pass (sae_bias_act_bwd_f32); no autograd."""
    lib = hip_lib.get()
    grad_output = grad_output.contiguous()
    lib.check(grad_output, out)
    step, size = _geometry(out)
    grad_input = torch.empty_like(grad_output)
    grad_bias = torch.empty(size, dtype=out.dtype, device=out.device)
    n_ws = lib.query("bias_act_bwd_workspace", out.numel(), step, size)
    ws = torch.empty(max(n_ws, 1), dtype=out.dtype, device=out.device)
    lib.call("bias_act_bwd_f32", grad_output.data_ptr(), out.data_ptr(), grad_input.data_ptr(),
             grad_bias.data_ptr(), ws.data_ptr(), n_ws, out.numel(), step, size, float(negative_slope),
             float(scale), lib.stream(out))
    return grad_input, grad_bias


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.set_materialize_grads(False)   # an undefined gradient skips the kernels instead of running them on zeros
        grad_input, grad_bias = bias_act_bwd_raw(grad_output, out, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        if gradgrad_input is None and gradgrad_bias is None:
            return None, None, None, None
        out, = ctx.saved_tensors
        negative_slope, scale = ctx.cfg
        # d(grad_input)/d(grad_output) and d(grad_bias)/d(grad_output) share the same mask:
        # gradgrad_out = (out > 0 ? 1 : slope) * scale * (gradgrad_input + gradgrad_bias[c])
        if gradgrad_input is None:
            gradgrad_input = torch.zeros_like(out)
        gradgrad_out = _bias_act(gradgrad_input, gradgrad_bias, out, 1, negative_slope, scale)
        return gradgrad_out, None, None, None


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
