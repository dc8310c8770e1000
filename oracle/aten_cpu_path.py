"""ATen-on-CPU restatement of the reference's CPU code path for the image discriminator.

TEST / BASELINE INFRASTRUCTURE ONLY (same rule as sae_oracle.c): only tests/ and bench.py's ``cpu_baseline`` leg
import this file.  It is what the reference executes when ``util.is_custom_kernel_supported`` is False
(util/util.py:432-436), i.e. on a CPU:

  * ``upfirdn2d_native``  (models/networks/stylegan2_op/upfirdn2d.py:162-222): zero-insert, F.pad (negative pads
    crop), F.conv2d with the flipped taps over a (N*C, 1, H, W) view, strided slice;
  * ``fused_leaky_relu``'s fallback (stylegan2_op/fused_act.py:93-96): ``F.leaky_relu(x + bias, 0.2) * sqrt(2)``;
  * ``EqualConv2d`` / ``EqualLinear`` (stylegan2_layers.py:115-195): F.conv2d / F.linear with ``weight * scale``;
  * ``ConvLayer`` / ``ResBlock`` / ``Discriminator`` (stylegan2_layers.py:612-763) composed from those.

Parity pin: tests/test_dropin_train.py::test_aten_cpu_path_matches_reference_discriminator runs the reference's
own Discriminator (imported from /root/reference in the build container) on the same weights and input.

bench.py times ``DiscriminatorCPU`` forward + backward on the host cores as the reported CPU baseline: the
discriminator is the largest consumer of the train step and runs through exactly the ATen
kernels (MKLDNN convolution, its two backward kernels, elementwise ops) that the rest of the reference's CPU path
uses."""
import math

import torch
import torch.nn.functional as F


def make_kernel(taps):
    k = torch.tensor(taps, dtype=torch.float32)
    k = torch.outer(k, k)
    return k / k.sum()


def upfirdn2d_native(x, kernel, up=1, down=1, pad=(0, 0)):
    n, c, h, w = x.shape
    v = x.reshape(n * c, 1, h, w)
    if up > 1:                                   # zero insertion (upfirdn2d.py:171-174)
        z = v.new_zeros(n * c, 1, h, up, w, up)
        z[:, :, :, 0, :, 0] = v
        v = z.reshape(n * c, 1, h * up, w * up)
    p0, p1 = pad
    v = F.pad(v, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    v = v[:, :, max(-p0, 0):v.shape[2] - max(-p1, 0), max(-p0, 0):v.shape[3] - max(-p1, 0)]
    v = F.conv2d(v, torch.flip(kernel, [0, 1])[None, None])
    v = v[:, :, ::down, ::down]
    return v.reshape(n, c, v.shape[2], v.shape[3])


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if bias is not None:
        x = x + bias.view(1, -1, *([1] * (x.dim() - 2)))
    return F.leaky_relu(x, negative_slope) * scale


class ConvLayerCPU(torch.nn.Module):
    def __init__(self, cin, cout, k, downsample=False, activate=True, bias=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin, k, k))
        self.scale = 1.0 / math.sqrt(cin * k * k)
        self.k, self.downsample, self.activate = k, downsample, activate
        self.bias = torch.nn.Parameter(torch.zeros(cout)) if bias else None
        self.register_buffer("blur", make_kernel([1, 3, 3, 1]))

    def forward(self, x):
        if self.downsample:
            p = (4 - 2) + (self.k - 1)
            x = upfirdn2d_native(x, self.blur, pad=((p + 1) // 2, p // 2))
            x = F.conv2d(x, self.weight * self.scale, stride=2)
        else:
            x = F.conv2d(x, self.weight * self.scale, padding=self.k // 2)
        if self.activate:
            return fused_leaky_relu(x, self.bias)
        return x if self.bias is None else x + self.bias.view(1, -1, 1, 1)

    def flops(self, n, h, w):
        """(forward FLOPs, output h, output w) for an input of n x cin x h x w"""
        cout, cin, k, _ = self.weight.shape
        oh, ow = (h // 2, w // 2) if self.downsample else (h, w)
        return 2.0 * n * cout * oh * ow * cin * k * k, oh, ow


class ResBlockCPU(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = ConvLayerCPU(cin, cin, 3)
        self.conv2 = ConvLayerCPU(cin, cout, 3, downsample=True)
        self.skip = ConvLayerCPU(cin, cout, 1, downsample=True, activate=False, bias=False)

    def forward(self, x):
        return (self.conv2(self.conv1(x)) + self.skip(x)) / math.sqrt(2)


class _HeadCPU(torch.nn.Module):
    """final_linear: EqualLinear(8192 -> 512, fused_lrelu) -> EqualLinear(512 -> 1)  (stylegan2_layers.py:736-739)"""

    def __init__(self, ch):
        super().__init__()
        self.lin1_w = torch.nn.Parameter(torch.randn(ch, ch * 16))
        self.lin1_b = torch.nn.Parameter(torch.zeros(ch))
        self.lin2_w = torch.nn.Parameter(torch.randn(1, ch))
        self.lin2_b = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        x = fused_leaky_relu(F.linear(x, self.lin1_w * (1.0 / math.sqrt(self.lin1_w.shape[1]))), self.lin1_b)
        return F.linear(x, self.lin2_w * (1.0 / math.sqrt(self.lin2_w.shape[1])), self.lin2_b)


class DiscriminatorCPU(torch.nn.Module):
    """stylegan2_layers.py:696-763 (minibatch-stddev commented out in the reference, :746-753)."""

    def __init__(self, size=256, channel_multiplier=2):
        super().__init__()
        cm = channel_multiplier
        ch = {4: 512, 8: 512, 16: min(512, int(512 * cm)), 32: min(512, int(512 * cm)), 64: int(256 * cm), 128: int(128 * cm),
              256: int(64 * cm), 512: int(32 * cm), 1024: int(16 * cm)}
        log_size = int(round(math.log2(size)))
        self.stem = ConvLayerCPU(3, ch[2 ** log_size], 1)
        blocks, cin = [], ch[2 ** log_size]
        for i in range(log_size, 2, -1):
            blocks.append(ResBlockCPU(cin, ch[2 ** (i - 1)]))
            cin = ch[2 ** (i - 1)]
        self.blocks = torch.nn.ModuleList(blocks)
        self.final_conv = ConvLayerCPU(cin, ch[4], 3)
        self.head = _HeadCPU(ch[4])

    def forward(self, x):
        x = self.stem(x)
        for b in self.blocks:
            x = b(x)
        return self.head(self.final_conv(x).flatten(1))

    def train_flops(self, n, size):
        """conv / linear FLOPs of forward + backward with trainable weights and an input that needs no gradient
        (the D(real) pass of a discriminator step): 3 x forward, minus the first layer's data gradient."""
        f, h, w = self.stem.flops(n, size, size)
        total = 2.0 * f                                  # stem: forward + wgrad only
        for b in self.blocks:
            f1, _, _ = b.conv1.flops(n, h, w)
            f2, oh, ow = b.conv2.flops(n, h, w)
            f3, _, _ = b.skip.flops(n, h, w)
            total += 3.0 * (f1 + f2 + f3)
            h, w = oh, ow
        total += 3.0 * self.final_conv.flops(n, h, w)[0]
        total += 3.0 * 2.0 * n * (self.head.lin1_w.numel() + self.head.lin2_w.numel())
        return total


# ---- one upsampling block of the generator (models/networks/generator.py:39-53) on the same ATen path --------------------
class ModulatedConvCPU(torch.nn.Module):
    """ModulatedConv2d with a [N, style_dim] style (stylegan2_layers.py:209-325, `new_demodulation` branch :279-287): the
    style modulates the INPUT (normalised over channels when demodulating), the weight is shared by the batch, repeated
    and demodulated per output channel, and the conv runs grouped over the batch (:300-321) -- restated literally,
    including the grouped conv_transpose2d + Blur of the upsampling form (:302-309)."""

    def __init__(self, cin, cout, k, styledim, upsample=False):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(1, cout, cin, k, k))
        self.mod_weight = torch.nn.Parameter(torch.randn(cin, styledim))      # modulation = EqualLinear(style_dim, cin, bias_init=1)
        self.mod_bias = torch.nn.Parameter(torch.ones(cin))
        self.cin, self.cout, self.k, self.upsample = cin, cout, k, upsample
        self.scale = 1.0 / math.sqrt(cin * k * k)
        self.mod_scale = 1.0 / math.sqrt(styledim)
        self.register_buffer("blur", make_kernel([1, 3, 3, 1]) * 4)         # Blur(..., upsample_factor=2): taps * factor^2

    def forward(self, x, style):
        b, cin, h, w = x.shape
        s = F.linear(style.view(b, -1), self.mod_weight * self.mod_scale, bias=self.mod_bias * 1.0)
        s = s.view(b, cin, 1, 1)
        s = s * torch.rsqrt(s.pow(2).mean([1], keepdim=True) + 1e-8)
        x = x * s
        weight = (self.scale * self.weight).repeat(b, 1, 1, 1, 1)
        demod = torch.rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8)
        weight = weight * demod.view(b, self.cout, 1, 1, 1)
        weight = weight.view(b * self.cout, cin, self.k, self.k)
        if self.upsample:
            x = x.view(1, b * cin, h, w)
            weight = weight.view(b, self.cout, cin, self.k, self.k).transpose(1, 2).reshape(b * cin, self.cout, self.k, self.k)
            out = F.conv_transpose2d(x, weight, padding=0, stride=2, groups=b)
            out = out.view(b, self.cout, out.shape[2], out.shape[3])
            p = (4 - 2) - (self.k - 1)
            return upfirdn2d_native(out, self.blur, pad=((p + 1) // 2 + 1, p // 2 + 1))
        x = x.view(1, b * cin, h, w)
        out = F.conv2d(x, weight, padding=self.k // 2, groups=b)
        return out.view(b, self.cout, out.shape[2], out.shape[3])


class _Holder(torch.nn.Module):
    def __init__(self, name, value):
        super().__init__()
        setattr(self, name, torch.nn.Parameter(value))


class StyledConvCPU(torch.nn.Module):
    """StyledConv (stylegan2_layers.py:367-405): modulated conv -> image + weight * noise (:351) -> FusedLeakyReLU."""

    def __init__(self, cin, cout, k, styledim, upsample=False):
        super().__init__()
        self.conv = ModulatedConvCPU(cin, cout, k, styledim, upsample=upsample)
        self.noise = _Holder("weight", torch.zeros(1))        # child modules, so that parameters() keeps the reference's order
        self.activate = _Holder("bias", torch.zeros(cout))

    def forward(self, x, style, noise):
        out = self.conv(x, style)
        out = out + self.noise.weight * noise
        return fused_leaky_relu(out, self.activate.bias)


class UpsamplingResnetBlockCPU(torch.nn.Module):
    """generator.py:39-53 with use_noise=True and inch != outch; parameters in the reference module's own order."""

    def __init__(self, inch, outch, styledim):
        super().__init__()
        self.conv1 = StyledConvCPU(inch, outch, 3, styledim, upsample=True)
        self.conv2 = StyledConvCPU(outch, outch, 3, styledim, upsample=False)
        self.skip = ConvLayerCPU(inch, outch, 1, activate=True, bias=True)

    def forward(self, x, style, noise1, noise2):
        skip = F.interpolate(self.skip(x), scale_factor=2, mode="bilinear", align_corners=False)
        res = self.conv2(self.conv1(x, style, noise1), style, noise2)
        return (skip + res) / math.sqrt(2)

