"""Layer library of the Swapping-Autoencoder networks on the MI355X operators.

Drop-in for the reference's ``models.networks.stylegan2_layers`` (itself derived from
rosinality/stylegan2-pytorch): same class names, constructor signatures, child-module and
parameter names (they are checkpoint keys) and the same forward semantics, cited
per class below — but every convolution, linear map, FIR blur and bias+activation runs on the
hand-written gfx950 kernels of ``stylegan2_op`` instead of ATen/cuDNN + CUDA extensions.

Not carried over (unused by the Swapping-Autoencoder networks): ``PixelNorm``,
``ConstantInput`` and the original StyleGAN2 ``Generator``.
"""
import math
import os
from collections import OrderedDict

import torch
from torch import nn
from torch.nn import functional as F

from . import hip_lib, rng
from .stylegan2_op import (FusedLeakyReLU, ReflectionPad2d, add_scale, blur_noise_bias_act, conv2d, conv2d_bias_act, conv_transpose2d,
                           fusable, fused_leaky_relu, l2_normalize, linear, modulated_conv2d, noise_bias_act, plane_scale,
                           reflect_pad, styled_modulated_conv2d, upfirdn2d)


# SAE_MODCONV_FUSED=0 (debug / A-B measurements): ModulatedConv2d takes the two-step path (x * s, then a plain conv)
_FUSED_MODCONV = os.environ.get("SAE_MODCONV_FUSED", "1") != "0"
# SAE_RESBLOCK_FUSED=0 (debug / A-B measurements): ResBlock runs module by module instead of as one autograd node
_FUSED_RESBLOCK = os.environ.get("SAE_RESBLOCK_FUSED", "1") != "0"
# SAE_STYLED_FUSED=0 (A/B): StyledConv runs conv, then noise + bias + activation, instead of the one-kernel form
_FUSED_STYLED = os.environ.get("SAE_STYLED_FUSED", "1") != "0"


def make_kernel(k):
    """Normalised FIR taps; a 1-D list becomes its outer product (stylegan2_layers.py:27-35)."""
    taps = torch.as_tensor(k, dtype=torch.float32)
    if taps.dim() == 1:
        taps = torch.outer(taps, taps)
    return taps / taps.sum()


def _split_pad(total):
    return (total + 1) // 2, total // 2


class Upsample(nn.Module):
    """Zero-insertion x``factor`` + FIR (stylegan2_layers.py:38-56).  Off the train path (API parity)."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    """FIR + decimation by ``factor`` (stylegan2_layers.py:59-87).  Off the train path (API parity)."""

    def __init__(self, kernel, factor=2, pad=None, reflection_pad=False):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel))
        self.reflection = reflection_pad
        self.pad = _split_pad(self.kernel.shape[0] - factor if pad is None else pad)

    def forward(self, input):
        pad = self.pad
        if self.reflection:
            input = reflect_pad(input, (pad[0], pad[1], pad[0], pad[1]))
            pad = (0, 0)
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=pad)


class Blur(nn.Module):
    """FIR blur with explicit pads, optionally after a reflection pad (stylegan2_layers.py:90-112)."""

    def __init__(self, kernel, pad, upsample_factor=1, reflection_pad=False):
        super().__init__()
        taps = make_kernel(kernel)
        if upsample_factor > 1:
            taps = taps * (upsample_factor ** 2)
        self.register_buffer("kernel", taps)
        self.pad = pad
        self.reflection = reflection_pad
        if reflection_pad:
            self.reflection_pad = ReflectionPad2d((pad[0], pad[1], pad[0], pad[1]))
            self.pad = (0, 0)

    def forward(self, input):
        if self.reflection:
            input = self.reflection_pad(input)
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """Equalised-lr conv: conv2d(x, weight * scale) + bias (stylegan2_layers.py:115-150); the scale
    is applied inside the MFMA kernel's weight staging."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True, lr_mul=1.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2) * lr_mul
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        return conv2d(input, self.weight, bias=self.bias, stride=self.stride, padding=self.padding,
                      alpha=self.scale)

    def __repr__(self):
        o, i, k, _ = self.weight.shape
        return "%s(%d, %d, %d, stride=%d, padding=%d)" % (type(self).__name__, i, o, k, self.stride, self.padding)


class EqualLinear(nn.Module):
    """Equalised-lr linear layer, optionally followed by the fused bias + leaky-ReLU
    (stylegan2_layers.py:153-195).  4-D input is treated as a 1x1 convolution (:175,:182)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def _matmul(self, input, bias):
        if input.dim() > 2:
            return conv2d(input, self.weight[:, :, None, None], bias=bias, alpha=self.scale)
        return linear(input, self.weight, bias=bias, alpha=self.scale)

    def forward(self, input):
        bias = self.bias * self.lr_mul if self.bias is not None else None
        if self.activation:
            return fused_leaky_relu(self._matmul(input, None), bias)
        return self._matmul(input, bias)

    def __repr__(self):
        return "%s(%d, %d)" % (type(self).__name__, self.weight.shape[1], self.weight.shape[0])


class ScaledLeakyReLU(nn.Module):
    """leaky_relu(x) * sqrt(2) (stylegan2_layers.py:198-207): the K2 kernel without a bias."""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return fused_leaky_relu(input, None, self.negative_slope, math.sqrt(2))


class ModulatedConv2d(nn.Module):
    """Style-modulated conv (stylegan2_layers.py:210-325) in its dense-equivalent form.

    With ``new_demodulation`` (always True in the reference, :258) the style scales the INPUT per
    sample (:280-286) and the demodulation factor is computed from the un-modulated weight
    (:290-292), so every sample shares one weight: ``conv2d(x * s, W * scale * demod)``.  The
    reference materialises ``batch`` copies of the weight and runs a groups=batch conv
    (:287,:299-321); here batch is just more pixels in the GEMM N dimension.  Verified equal to the
    reference module to fp32 round-off in tests/test_reference_parity.py.
    """

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=_split_pad(p))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self.new_demodulation = True
        self._projected_style = None   # set per pass by StyleGAN2ResnetGenerator (batched projection)

    def __repr__(self):
        return "%s(%d, %d, %d, upsample=%s, downsample=%s)" % (
            type(self).__name__, self.in_channel, self.out_channel, self.kernel_size, self.upsample, self.downsample)

    def _input_scale(self, input, style):
        if style.dim() > 2:   # spatially varying style (:269-277); off the train path
            style = F.interpolate(style, size=input.shape[2:], mode="bilinear", align_corners=False)
            s = self.modulation(style)
            if self.demodulate:
                s = s * torch.rsqrt(s.pow(2).mean([1], keepdim=True) + 1e-8)
            return s
        if self._projected_style is not None:     # the generator projected all styles in one GEMM
            s = self._projected_style
        else:
            s = self.modulation(style.view(input.shape[0], -1))
        if self.demodulate:
            s = s * torch.rsqrt(s.pow(2).mean([1], keepdim=True) + 1e-8)
        return s        # [N, C]: broadcast over the plane by the caller

    def _weight(self):
        w = self.weight[0] * self.scale
        if self.demodulate:
            w = w * torch.rsqrt(w.pow(2).sum([1, 2, 3], keepdim=True) + 1e-8)
        return w

    def _demod(self):
        """rsqrt(sum_{i,kh,kw} (W * scale)^2 + eps) per output channel (:290-292), from the un-modulated weight"""
        # evaluated exactly as the reference does (scale first, then square, sum, rsqrt): the conv kernels then see
        # bit-identical operands on the fused and on the two-step path (a leaky-ReLU sitting within 1e-7 of zero
        # downstream turns any other rounding into a visibly different gradient)
        w = self.weight[0] * self.scale
        return torch.rsqrt(w.pow(2).sum(dim=(1, 2, 3)) + 1e-8)

    def _fused_operands(self, input, style):
        """(style factor [N, C], output scale) for the one-kernel path, or None when it does not apply."""
        fused = (not self.downsample and input.dtype == torch.float32 and _FUSED_MODCONV and hip_lib.get_conv_math() == "f32"
                 and (style.dim() == 2 or self._projected_style is not None))
        if not fused:
            return None
        # style normalisation as ONE kernel forward, one backward (csrc/glue.hip l2_normalize):
        #   s * rsqrt(mean(s^2) + eps) = sqrt(C) * s * rsqrt(sum(s^2) + C eps)
        # the constant sqrt(C) rides in the conv's output scale (the conv is linear in s); the demodulation factor is
        # computed and differentiated inside the conv node (weight_demod_*).  ~40 ATen launches per conv and
        # forward + backward pass become 4.
        s = self._projected_style if self._projected_style is not None else self.modulation(style.view(input.shape[0], -1))
        out_scale = 1.0
        if self.demodulate:
            c = s.shape[1]
            s = l2_normalize(s, c * 1e-8)
            out_scale = math.sqrt(c)
        return s, out_scale

    def forward(self, input, style):
        ops = self._fused_operands(input, style)
        if ops is not None:
            s, out_scale = ops
            out = modulated_conv2d(input, s, self.weight.view(self.weight.shape[1:]), None, padding=self.padding, alpha=self.scale,
                                   transposed=self.upsample, demod_eps=self.eps if self.demodulate else None, out_scale=out_scale)
            return self.blur(out) if self.upsample else out
        s = self._input_scale(input, style)
        if (s.dim() == 2 and not self.downsample and input.dtype == torch.float32 and _FUSED_MODCONV
                and hip_lib.get_conv_math() == "f32"):
            # ONE kernel per operation: style factor folded into the conv's operand staging, demodulation into its
            # weight re-layout (stylegan2_op.conv2d_gemm.ModulatedConv).  Under the bf16x6 arithmetic the operand
            # staging does not take activation factors yet: the two-step path below runs there.
            # LIMITATION: this fused node is differentiable ONCE (ModulatedConv.backward is @once_differentiable); the
            # two-step path below is closed under differentiation.  The train step never differentiates the generator
            # twice (the R1 penalties touch D and Dpatch only); a second-order gradient through G needs
            # SAE_MODCONV_FUSED=0 (or SAE_CONV_MATH=bf16x6), otherwise autograd raises.
            out = modulated_conv2d(input, s, self.weight[0], self._demod() if self.demodulate else None,
                                   padding=self.padding, alpha=self.scale, transposed=self.upsample)
            return self.blur(out) if self.upsample else out
        if s.dim() == 2:
            # x * s[:, :, None, None]; its backward (g * s and sum_hw g * x) is one fused pass
            x = plane_scale(input, s) if fusable(input) else input * s[:, :, None, None]
        else:
            x = input * s
        w = self._weight()
        if self.upsample:
            return self.blur(conv_transpose2d(x, w, stride=2))
        if self.downsample:
            return conv2d(self.blur(x), w, stride=2, padding=0)
        return conv2d(x, w, padding=self.padding)


class NoiseInjection(nn.Module):
    """image + weight * noise with a fresh N(0,1) map per call unless a noise is passed or fixed
    (stylegan2_layers.py:328-351).  ``image_size`` / ``fixed_noise`` are read by
    BaseNetwork.fix_and_gather_noise_parameters (base_network.py:42-51)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))
        self.fixed_noise = None
        self.image_size = None

    def resolve(self, image, noise=None):
        """The noise map this call uses (:343-350): the fixed one, the one passed in, or a fresh N(0,1) map."""
        if self.image_size is None:
            self.image_size = image.shape
        if self.fixed_noise is not None:
            noise = self.fixed_noise
            if noise.shape[2:] != image.shape[2:]:
                noise = F.interpolate(noise, image.shape[2:], mode="nearest")
        elif noise is None:
            noise = rng.randn_like_image(image)
        return noise

    def forward(self, image, noise=None):
        return image + self.weight * self.resolve(image, noise)


class StyledConv(nn.Module):
    """ModulatedConv2d -> noise -> bias + leaky-ReLU (stylegan2_layers.py:367-405)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, use_noise=True, lr_mul=1.0):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.use_noise = use_noise
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, act_ticket=None, input_ticket=None, grad_scale_ticket=None):
        """act_ticket / input_ticket (stylegan2_op.modulate.ActTicket): set by the generator blocks only -- this layer's output
        has ONE consumer (the next StyledConv of the block), whose backward then also runs this layer's activation backward."""
        conv = self.conv
        out = None
        if self.use_noise and _FUSED_STYLED and not conv.upsample and not conv.downsample and input.dim() == 4:
            ops = conv._fused_operands(input, style)
            if ops is not None:
                # conv -> noise -> bias + leaky-ReLU as ONE kernel (the noise map rides in the conv's epilogue): the noise is
                # drawn first, from a memory-less probe of the output's shape -- same stream of draws as the module path
                n, _, h, w = input.shape
                oh, ow = h + 2 * conv.padding - conv.kernel_size + 1, w + 2 * conv.padding - conv.kernel_size + 1
                z = self.noise.resolve(input.new_empty(1).expand(n, conv.out_channel, oh, ow), noise)
                if tuple(z.shape) == (n, 1, oh, ow) and not z.requires_grad and z.dtype == torch.float32:
                    s, out_scale = ops
                    return styled_modulated_conv2d(input, s, conv.weight.view(conv.weight.shape[1:]), z, self.noise.weight,
                                                   self.activate.bias, padding=conv.padding, alpha=conv.scale,
                                                   demod_eps=conv.eps if conv.demodulate else None, out_scale=out_scale,
                                                   negative_slope=self.activate.negative_slope, scale=self.activate.scale,
                                                   act_ticket=act_ticket, input_ticket=input_ticket,
                                                   grad_scale_ticket=grad_scale_ticket)
                noise = z       # drawn already: the module path below must not draw again
        elif (self.use_noise and _FUSED_STYLED and conv.upsample and input.dim() == 4 and not conv.blur.reflection
              and max(conv.blur.kernel.shape) <= 4):
            ops = conv._fused_operands(input, style)
            if ops is not None:
                # the upsampling form: transposed conv, then blur -> noise -> bias + leaky-ReLU as ONE K1 kernel
                s, out_scale = ops
                up = modulated_conv2d(input, s, conv.weight.view(conv.weight.shape[1:]), None, padding=conv.padding,
                                      alpha=conv.scale, transposed=True, demod_eps=conv.eps if conv.demodulate else None,
                                      out_scale=out_scale)
                n, c, h, w = up.shape
                kh, kw = conv.blur.kernel.shape
                oh, ow = h + sum(conv.blur.pad) - kh + 1, w + sum(conv.blur.pad) - kw + 1
                z = self.noise.resolve(up.new_empty(1).expand(n, c, oh, ow), noise)
                if tuple(z.shape) == (n, 1, oh, ow) and not z.requires_grad and z.dtype == torch.float32:
                    return blur_noise_bias_act(up, conv.blur.kernel, conv.blur.pad, z, self.noise.weight, self.activate.bias,
                                               self.activate.negative_slope, self.activate.scale, act_ticket=act_ticket)
                out, noise = conv.blur(up), z     # an unusual noise map: blur, then the module path's tail with the drawn map
        if out is None:
            out = self.conv(input, style)
        if self.use_noise:
            z = self.noise.resolve(out, noise)
            if fusable(out) and z.shape == (out.shape[0], 1) + out.shape[2:] and not z.requires_grad:
                # noise + bias + leaky-ReLU in one pass (and one backward pass for all three gradients).  A noise map
                # that itself asks for a gradient (an optimised / projected `fixed_noise` parameter,
                # base_network.py:42-51) takes the unfused path below, which differentiates it
                return noise_bias_act(out, z, self.noise.weight, self.activate.bias, self.activate.negative_slope,
                                      self.activate.scale)
            out = out + self.noise.weight * z
        return self.activate(out)


class ToRGB(nn.Module):
    """1x1 modulated conv without demodulation + bias (+ upsampled skip) (stylegan2_layers.py:408-427)."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None):
        out = self.conv(input, style) + self.bias
        if skip is not None:
            out = out + self.upsample(skip)
        return out


class ConvLayer(nn.Sequential):
    """[Blur | RefPad] -> Conv -> [Act] with the reference's child names (stylegan2_layers.py:612-668)."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True, pad=None, reflection_pad=False):
        layers = []
        if downsample:
            factor = 2
            if pad is None:
                pad = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(("Blur", Blur(blur_kernel, pad=_split_pad(pad), reflection_pad=reflection_pad)))
            stride = 2
            self.padding = 0
        else:
            stride = 1
            self.padding = kernel_size // 2 if pad is None else pad
            if reflection_pad:
                layers.append(("RefPad", ReflectionPad2d(self.padding)))
                self.padding = 0
        layers.append(("Conv", EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                           bias=bias and not activate)))
        if activate:
            layers.append(("Act", FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2)))
        super().__init__(OrderedDict(layers))
        self._activated = activate

    def _stem_config(self):
        """StemConfig when this layer is a plain activated stride-1 conv (no blur, no reflection pad, FusedLeakyReLU with a
        bias, size-preserving padding), else None."""
        if not _FUSED_RESBLOCK or len(self._modules) != 2 or not isinstance(self._modules.get("Act"), FusedLeakyReLU):
            return None
        conv = self.Conv
        k = conv.weight.shape[2]
        if conv.stride != 1 or k not in (1, 3) or conv.padding != k // 2 or conv.bias is not None:
            return None
        from .stylegan2_op.resblock import StemConfig
        return StemConfig(k, conv.padding, conv.scale, self.Act.negative_slope, self.Act.scale)

    def forward(self, input):
        """Same result as running the children in order; the Conv -> Act pair is ONE kernel (the
        bias + leaky-ReLU sits in the conv's epilogue)."""
        if "Blur" in self._modules and self.Conv.weight.shape[2] == 1 and self.Conv.stride == 2:
            # skip path of a downsampling ResBlock: blur at full resolution, then a 1x1 conv that
            # reads every other pixel (:627-634,:647-655).  Decimating inside upfirdn2d (down=2)
            # computes only the pixels the conv reads: a quarter of the FIR work and traffic, and
            # the conv becomes stride 1 on the small image.  Same values (same taps, same sums).
            blur, conv = self.Blur, self.Conv
            if blur.reflection:
                input = blur.reflection_pad(input)
            input = upfirdn2d(input, blur.kernel, up=1, down=2, pad=blur.pad)
            if not self._activated:
                return conv2d(input, conv.weight, bias=conv.bias, stride=1, padding=0, alpha=conv.scale)
            act = self.Act
            bias = act.bias if isinstance(act, FusedLeakyReLU) else None
            act_scale = act.scale if isinstance(act, FusedLeakyReLU) else math.sqrt(2)
            return conv2d_bias_act(input, conv.weight, bias, stride=1, padding=0, alpha=conv.scale,
                                   negative_slope=act.negative_slope, scale=act_scale)
        if not self._activated:
            return super().forward(input)
        for name, child in self.named_children():
            if name in ("Blur", "RefPad"):
                input = child(input)
        conv, act = self.Conv, self.Act
        bias = act.bias if isinstance(act, FusedLeakyReLU) else None
        act_scale = act.scale if isinstance(act, FusedLeakyReLU) else math.sqrt(2)
        return conv2d_bias_act(input, conv.weight, bias, stride=conv.stride, padding=conv.padding, alpha=conv.scale,
                               negative_slope=act.negative_slope, scale=act_scale)


class ResBlock(nn.Module):
    """(conv2(conv1(x)) + skip(x)) / sqrt(2) (stylegan2_layers.py:672-693)."""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1], reflection_pad=False, pad=None,
                 downsample=True):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3, reflection_pad=reflection_pad, pad=pad)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=downsample, blur_kernel=blur_kernel,
                               reflection_pad=reflection_pad, pad=pad)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=downsample, blur_kernel=blur_kernel,
                              activate=False, bias=False)

        self._fused_cfg = None

    def _fused_config(self):
        """ResBlockConfig of the single-node path (stylegan2_op/resblock.py), or False when this block is not the plain
        downsampling block of D / Dpatch (reflection-padded encoder blocks, the non-downsampling block, other taps)."""
        if self._fused_cfg is None:
            c1, c2, sk = self.conv1, self.conv2, self.skip
            ok = (_FUSED_RESBLOCK and "Blur" in c2._modules and "Blur" in sk._modules and "RefPad" not in c1._modules
                  and not c2.Blur.reflection and not sk.Blur.reflection
                  and isinstance(c1._modules.get("Act"), FusedLeakyReLU) and isinstance(c2._modules.get("Act"), FusedLeakyReLU)
                  and "Act" not in sk._modules and sk.Conv.bias is None
                  and tuple(c1.Conv.weight.shape[2:]) == (3, 3) and c1.Conv.stride == 1 and c1.Conv.padding == 1
                  and tuple(c2.Conv.weight.shape[2:]) == (3, 3) and c2.Conv.stride == 2 and c2.Conv.padding == 0
                  and tuple(sk.Conv.weight.shape[2:]) == (1, 1) and sk.Conv.stride == 2
                  and max(c2.Blur.kernel.shape) <= 4 and max(sk.Blur.kernel.shape) <= 4)
            if ok:
                from .stylegan2_op.resblock import ResBlockConfig
                self._fused_cfg = ResBlockConfig(c2.Blur.kernel, c2.Blur.pad, sk.Blur.kernel, sk.Blur.pad, c1.Conv.scale,
                                                 c2.Conv.scale, sk.Conv.scale, c1.Act.negative_slope, c1.Act.scale,
                                                 c2.Act.negative_slope, c2.Act.scale, 1.0 / math.sqrt(2))
            else:
                self._fused_cfg = False
        return self._fused_cfg

    def forward(self, input, stem=None):
        """stem: an activated stride-1 ConvLayer whose output is this block's input (run_sequence): it is then evaluated
        inside the block's autograd node, `input` being the stem's input."""
        cfg = self._fused_config()
        if cfg and input.shape[2] % 2 == 0 and input.shape[3] % 2 == 0:
            from .stylegan2_op.resblock import resblock
            # the Blur buffers may have moved (module.to(device)) since the config was made
            cfg.taps2, cfg.taps_s = self.conv2.Blur.kernel, self.skip.Blur.kernel
            st = None if stem is None else (stem.Conv.weight, stem.Act.bias, stem._stem_config())
            return resblock(input, self.conv1.Conv.weight, self.conv1.Act.bias, self.conv2.Conv.weight, self.conv2.Act.bias,
                            self.skip.Conv.weight, cfg, stem=st)
        if stem is not None:
            input = stem(input)
        return add_scale(self.conv2(self.conv1(input)), self.skip(input), 1.0 / math.sqrt(2))


def run_sequence(seq, x):
    """``seq(x)`` for an nn.Sequential of ConvLayers / ResBlocks (D's and Dpatch's ``convs``), except that an activated
    stride-1 ConvLayer directly in front of a single-node ResBlock is evaluated inside that node (its activation's
    backward then rides in the kernel that finishes the block's input gradient).  Same values either way."""
    mods = list(seq.children())
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if (isinstance(m, ConvLayer) and isinstance(nxt, ResBlock) and m._stem_config() and nxt._fused_config()
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.dim() == 4):
            x = nxt(x, stem=m)
            i += 2
        else:
            x = m(x)
            i += 1
    return x


class Discriminator(nn.Module):
    """StyleGAN2 residual discriminator without the minibatch-stddev layer, which the reference
    has commented out (stylegan2_layers.py:696-763)."""

    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        cm = channel_multiplier
        channels = {4: 512, 8: 512, 16: min(512, int(512 * cm)), 32: min(512, int(512 * cm)), 64: int(256 * cm),
                    128: int(128 * cm), 256: int(64 * cm), 512: int(32 * cm), 1024: int(16 * cm)}
        original_size = size
        log_size = int(round(math.log(size, 2)))
        size = 2 ** log_size
        in_channel = channels[size]
        convs = [("0", ConvLayer(3, in_channel, 1))]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            name = str(9 - i) if i <= 8 else "%dx%d" % (2 ** i, 2 ** i)
            convs.append((name, ResBlock(in_channel, out_channel, blur_kernel)))
            in_channel = out_channel
        self.convs = nn.Sequential(OrderedDict(convs))
        self.final_conv = ConvLayer(in_channel, channels[4], 3)
        side = int(4 * original_size / size)
        self.final_linear = nn.Sequential(
            EqualLinear(channels[4] * side * side, channels[4], activation="fused_lrelu"),
            EqualLinear(channels[4], 1),
        )

    def forward(self, input):
        out = self.final_conv(run_sequence(self.convs, input))
        return self.final_linear(out.view(out.shape[0], -1))

    def get_features(self, input):
        return self.final_conv(run_sequence(self.convs, input))
