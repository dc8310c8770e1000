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


class ActivationMasks:
    """Leaky-ReLU sign patterns of one run, for the mask-frozen double reference of the whole-network parity tests
    (tests/test_gpu_network_parity.py).  A pre-activation within rounding of zero takes the other branch in an fp32 run
    than in the double run, and the gradient of the network is discontinuous across that flip; evaluated with the fp32
    run's OWN sign pattern the double run is the exact derivative of the piecewise-linear function the fp32 run
    evaluated, a continuous function of inputs and weights, and a tolerance can be asserted on gradients directly.

      with ActivationMasks.record() as m:    run the restatement (any dtype): m.masks = sign patterns in call order
      with ActivationMasks.replay(m.masks):  the same restatement uses them, one per activation call, in call order
      with ActivationMasks.lookup(fn):       fn(pre_activation, activated) -> bool mask (or None = keep own sign):
                                             for a run whose patterns are recovered from elsewhere (another
                                             implementation's saved activations)
    ``flips`` counts, per activation call, the elements whose imposed sign differs from the run's own."""
    _active = None

    def __init__(self, mode, masks=None, fn=None):
        self.mode, self.masks, self.fn, self.flips, self._i = mode, ([] if masks is None else masks), fn, [], 0

    @classmethod
    def record(cls):
        return cls("record")

    @classmethod
    def replay(cls, masks):
        return cls("replay", masks=masks)

    @classmethod
    def lookup(cls, fn):
        return cls("lookup", fn=fn)

    def __enter__(self):
        assert ActivationMasks._active is None
        ActivationMasks._active = self
        return self

    def __exit__(self, *exc):
        ActivationMasks._active = None
        if self.mode == "replay" and exc[0] is None:
            assert self._i == len(self.masks), "replayed %d of %d activation masks" % (self._i, len(self.masks))
        return False

    def mask_for(self, pre, slope, scale):
        own = pre > 0
        if self.mode == "record":
            self.masks.append(own)
            return None
        if self.mode == "replay":
            m = self.masks[self._i]
            self._i += 1
        else:
            m = self.fn(pre.detach(), (torch.where(own, pre, pre * slope) * scale).detach())
            if m is None:
                self.flips.append(-1)
                return None
        assert m.shape == pre.shape, (m.shape, pre.shape)
        self.flips.append(int((m != own).sum()))
        return m


def _leaky_relu(x, negative_slope, scale):
    a = ActivationMasks._active
    m = a.mask_for(x, negative_slope, scale) if a is not None else None
    return (F.leaky_relu(x, negative_slope) if m is None else torch.where(m, x, x * negative_slope)) * scale


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if bias is not None:
        x = x + bias.view(1, -1, *([1] * (x.dim() - 2)))
    return _leaky_relu(x, negative_slope, scale)


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

    def __init__(self, cin, cout, k, styledim, upsample=False, demodulate=True):
        super().__init__()
        self.demodulate = demodulate
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
        if self.demodulate:
            s = s * torch.rsqrt(s.pow(2).mean([1], keepdim=True) + 1e-8)
        x = x * s
        weight = (self.scale * self.weight).repeat(b, 1, 1, 1, 1)
        if self.demodulate:
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


def noise_map(out):
    """A fresh N(0, 1) map for `out` ([b, c, h, w]).  On the CPU: drawn in place, as the reference does.  When the restatement
    is moved to another device / to double for a parity run (tests/test_gpu_step_parity.py), the draw still comes from the
    CPU generator as an fp32 map -- the stream the compared run is fed from -- and is then moved."""
    if out.device.type == "cpu" and out.dtype == torch.float32:
        return out.new_empty(out.shape[0], 1, out.shape[2], out.shape[3]).normal_()
    return torch.empty(out.shape[0], 1, out.shape[2], out.shape[3]).normal_().to(out)


class StyledConvCPU(torch.nn.Module):
    """StyledConv (stylegan2_layers.py:367-405): modulated conv -> image + weight * noise (:351) -> FusedLeakyReLU."""

    def __init__(self, cin, cout, k, styledim, upsample=False):
        super().__init__()
        self.conv = ModulatedConvCPU(cin, cout, k, styledim, upsample=upsample)
        self.noise = _Holder("weight", torch.zeros(1))        # child modules, so that parameters() keeps the reference's order
        self.activate = _Holder("bias", torch.zeros(cout))

    def forward(self, x, style, noise=None):
        out = self.conv(x, style)
        if noise is None and getattr(self, "fixed_noise", None) is not None:      # :343-347, the fixed map
            noise = self.fixed_noise.to(out.dtype)
        if noise is None:      # NoiseInjection.forward, stylegan2_layers.py:340-342: a fresh N(0, 1) map per call
            noise = noise_map(out)
        out = out + self.noise.weight * noise
        return fused_leaky_relu(out, self.activate.bias)


class UpsamplingResnetBlockCPU(torch.nn.Module):
    """generator.py:39-53 with use_noise=True; parameters in the reference module's own order."""

    def __init__(self, inch, outch, styledim):
        super().__init__()
        self.conv1 = StyledConvCPU(inch, outch, 3, styledim, upsample=True)
        self.conv2 = StyledConvCPU(outch, outch, 3, styledim, upsample=False)
        self.skip = ConvLayerCPU(inch, outch, 1, activate=True, bias=True) if inch != outch else None   # Identity, :48-49

    def forward(self, x, style, noise1=None, noise2=None):
        skip = F.interpolate(x if self.skip is None else self.skip(x), scale_factor=2, mode="bilinear", align_corners=False)
        res = self.conv2(self.conv1(x, style, noise1), style, noise2)
        return (skip + res) / math.sqrt(2)


# ---- the encoder and the generator of the reconstruction path (BASELINE.json config 1: 32 x 32, B = 4, one reconstruction
# forward + backward on the CPU) on the same ATen path; parameters in the reference modules' own order ----------------------
def normalize(v):
    """util/util.py:18-22"""
    return v * torch.rsqrt(torch.sum(v ** 2, dim=1, keepdim=True) + 1e-8)


class GenConvLayerCPU(torch.nn.Module):
    """ConvLayer in full (stylegan2_layers.py:612-668): optional Blur (with its own reflection padding, :90-112) before a
    stride-2 conv, or (reflection) padding before a stride-1 conv; conv bias only without activation; FusedLeakyReLU /
    ScaledLeakyReLU after."""

    def __init__(self, cin, cout, k, downsample=False, blur_kernel=(1, 3, 3, 1), bias=True, activate=True, pad=None,
                 reflection_pad=False):
        super().__init__()
        self.k, self.downsample, self.activate, self.reflection = k, downsample, activate, reflection_pad
        self.weight = torch.nn.Parameter(torch.randn(cout, cin, k, k))
        self.scale = 1.0 / math.sqrt(cin * k * k)
        self.conv_bias = torch.nn.Parameter(torch.zeros(cout)) if (bias and not activate) else None
        self.act_bias = torch.nn.Parameter(torch.zeros(cout)) if (bias and activate) else None
        if downsample:
            if pad is None:
                pad = (len(blur_kernel) - 2) + (k - 1)
            self.blur_pad = ((pad + 1) // 2, pad // 2)
            self.register_buffer("blur", make_kernel(list(blur_kernel)))
            self.padding = 0
        else:
            self.padding = k // 2 if pad is None else pad

    def forward(self, x):
        if self.downsample:
            p0, p1 = self.blur_pad
            if self.reflection:
                x = upfirdn2d_native(F.pad(x, (p0, p1, p0, p1), mode="reflect"), self.blur, pad=(0, 0))
            else:
                x = upfirdn2d_native(x, self.blur, pad=(p0, p1))
            x = F.conv2d(x, self.weight * self.scale, bias=self.conv_bias, stride=2, padding=0)
        else:
            padding = self.padding
            if self.reflection and padding > 0:
                x = F.pad(x, (padding,) * 4, mode="reflect")
                padding = 0
            x = F.conv2d(x, self.weight * self.scale, bias=self.conv_bias, stride=1, padding=padding)
        if self.activate:
            if self.act_bias is not None:
                return fused_leaky_relu(x, self.act_bias)
            return _leaky_relu(x, 0.2, math.sqrt(2))             # ScaledLeakyReLU, stylegan2_layers.py:198-207
        return x


class GenResBlockCPU(torch.nn.Module):
    """ResBlock (stylegan2_layers.py:672-693) with its blur taps / reflection padding options."""

    def __init__(self, cin, cout, blur_kernel=(1, 3, 3, 1), reflection_pad=False, downsample=True):
        super().__init__()
        self.conv1 = GenConvLayerCPU(cin, cin, 3, reflection_pad=reflection_pad)
        self.conv2 = GenConvLayerCPU(cin, cout, 3, downsample=downsample, blur_kernel=blur_kernel, reflection_pad=reflection_pad)
        self.skip = GenConvLayerCPU(cin, cout, 1, downsample=downsample, blur_kernel=blur_kernel, activate=False, bias=False)

    def forward(self, x):
        return (self.conv2(self.conv1(x)) + self.skip(x)) / math.sqrt(2)


class _LinearCPU(torch.nn.Module):
    """EqualLinear without activation (stylegan2_layers.py:152-190)."""

    def __init__(self, cin, cout, bias_init=0.0):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin))
        self.bias = torch.nn.Parameter(torch.full((cout,), float(bias_init)))
        self.scale = 1.0 / math.sqrt(cin)

    def forward(self, x):
        return F.linear(x, self.weight * self.scale, bias=self.bias * 1.0)


class EncoderCPU(torch.nn.Module):
    """StyleGAN2ResnetEncoder (models/networks/encoder.py:31-114), forward without feature extraction."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        n_sp, n_gl = opt.netE_num_downsampling_sp, opt.netE_num_downsampling_gl
        blur = (1, 2, 1) if opt.use_antialias else (1,)
        self.FromRGB = GenConvLayerCPU(3, self.nc(0), 1)
        self.DownToSpatialCode = torch.nn.ModuleList(
            [GenResBlockCPU(self.nc(i), self.nc(i + 1), blur, reflection_pad=True) for i in range(n_sp)])
        ch = self.nc(n_sp)
        self.ToSpatialCode = torch.nn.ModuleList([GenConvLayerCPU(ch, ch, 1, activate=True, bias=True),
                                                  GenConvLayerCPU(ch, opt.spatial_code_ch, 1, activate=False, bias=True)])
        self.DownToGlobalCode = torch.nn.ModuleList(
            [GenConvLayerCPU(self.nc(n_sp + i), self.nc(n_sp + i + 1), 3, blur_kernel=(1,), downsample=True, pad=0)
             for i in range(n_gl)])
        self.ToGlobalCode = _LinearCPU(self.nc(n_sp + n_gl), opt.global_code_ch)

    def nc(self, idx):
        nc = self.opt.netE_nc_steepness ** (5 + idx) * self.opt.netE_scale_capacity
        return round(min(self.opt.global_code_ch, int(round(nc))))

    def forward(self, x):
        x = self.FromRGB(x)
        for b in self.DownToSpatialCode:
            x = b(x)
        sp = x
        for c in self.ToSpatialCode:
            sp = c(sp)
        for c in self.DownToGlobalCode:
            x = c(x)
        gl = self.ToGlobalCode(x.mean(dim=(2, 3)))
        return normalize(sp), normalize(gl)


class ResolutionPreservingResnetBlockCPU(torch.nn.Module):
    """generator.py:23-36"""

    def __init__(self, inch, outch, styledim):
        super().__init__()
        self.conv1 = StyledConvCPU(inch, outch, 3, styledim)
        self.conv2 = StyledConvCPU(outch, outch, 3, styledim)
        self.skip = GenConvLayerCPU(inch, outch, 1, activate=False, bias=False) if inch != outch else None

    def forward(self, x, style):
        skip = x if self.skip is None else self.skip(x)
        return (skip + self.conv2(self.conv1(x, style), style)) / math.sqrt(2)


class ToRGBCPU(torch.nn.Module):
    """ToRGB with skip = None (stylegan2_layers.py:408-427): un-demodulated 1x1 modulated conv + bias."""

    def __init__(self, cin, styledim):
        super().__init__()
        self.bias = torch.nn.Parameter(torch.zeros(1, 3, 1, 1))
        self.conv = ModulatedConvCPU(cin, 3, 1, styledim, demodulate=False)

    def forward(self, x, style):
        return self.conv(x, style) + self.bias


class GeneratorCPU(torch.nn.Module):
    """StyleGAN2ResnetGenerator (models/networks/generator.py:70-186) with fresh noise maps per call."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        sd = opt.global_code_ch + opt.num_classes
        self.mod_scale = _LinearCPU(sd, opt.spatial_code_ch)          # GeneratorModulation: scale, bias (generator.py:56-67)
        self.mod_bias = _LinearCPU(sd, opt.spatial_code_ch)
        blocks, cin = [], opt.spatial_code_ch
        for i in range(opt.netG_num_base_resnet_layers):
            cout = max(opt.spatial_code_ch, round((i + 1) / opt.netG_num_base_resnet_layers * self.nf(0)))
            blocks.append(ResolutionPreservingResnetBlockCPU(cin, cout, sd))
            cin = cout
        self.head = torch.nn.ModuleList(blocks)
        ups = []
        for j in range(opt.netE_num_downsampling_sp):
            cout = self.nf(j + 1)
            ups.append(UpsamplingResnetBlockCPU(cin, cout, sd))
            cin = cout
        self.ups = torch.nn.ModuleList(ups)
        self.to_rgb = ToRGBCPU(cin, sd)

    def nf(self, num_up):
        ch = 128 * (2 ** (self.opt.netE_num_downsampling_sp - num_up))
        return int(min(512, ch) * self.opt.netG_scale_capacity)

    def forward(self, sp, gl):
        sp, gl = normalize(sp), normalize(gl)
        x = sp * (1 * self.mod_scale(gl)[:, :, None, None]) + self.mod_bias(gl)[:, :, None, None]
        for b in self.head:
            x = b(x, gl)
        for b in self.ups:
            x = b(x, gl)
        return self.to_rgb(x, gl)



# ---- the patch discriminator, the crop sampler, the losses and ONE full training iteration (a discriminator call and a
# generator call of the driver) on the same ATen path: bench.py --full-cpu-baseline; pinned to the reference's golden
# loss dictionaries in tests/test_network_parity_cpu.py --------------------------------------------------------------------------
class _ActLinearCPU(torch.nn.Module):
    """EqualLinear(activation='fused_lrelu') (stylegan2_layers.py:172-190): linear without bias, then fused_leaky_relu(out, bias)"""

    def __init__(self, cin, cout, activate):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin))
        self.bias = torch.nn.Parameter(torch.zeros(cout))
        self.scale, self.activate = 1.0 / math.sqrt(cin), activate

    def forward(self, x):
        if self.activate:
            return fused_leaky_relu(F.linear(x, self.weight * self.scale), self.bias * 1.0)
        return F.linear(x, self.weight * self.scale, bias=self.bias * 1.0)


class PatchDiscriminatorCPU(torch.nn.Module):
    """StyleGAN2PatchDiscriminator (models/networks/patch_discriminator.py:97-171)"""

    def __init__(self, opt):
        super().__init__()
        cm, max_nc, size = opt.netPatchD_scale_capacity, opt.netPatchD_max_nc, opt.patch_size
        channels = {4: min(max_nc, int(256 * cm)), 8: min(max_nc, int(128 * cm)), 16: min(max_nc, int(64 * cm)),
                    32: int(32 * cm), 64: int(16 * cm), 128: int(8 * cm), 256: int(4 * cm)}
        log_size = int(math.ceil(math.log(size, 2)))
        blur = (1, 3, 3, 1) if opt.use_antialias else (1,)
        ch = channels[2 ** log_size]
        convs = [GenConvLayerCPU(3, ch, 3)]
        for i in range(log_size, 2, -1):
            convs.append(GenResBlockCPU(ch, channels[2 ** (i - 1)], blur))
            ch = channels[2 ** (i - 1)]
        convs.append(GenResBlockCPU(ch, max_nc * 2, downsample=False))
        convs.append(GenConvLayerCPU(max_nc * 2, max_nc, 3, pad=0))
        self.convs = torch.nn.ModuleList(convs)
        self.pairlinear = torch.nn.ModuleList([_ActLinearCPU(channels[4] * 2 * 2 * 2, 2048, True), _ActLinearCPU(2048, 2048, True),
                                               _ActLinearCPU(2048, 1024, True), _ActLinearCPU(1024, 1, False)])

    def extract_features(self, patches, aggregate=False):
        b, t = patches.shape[:2]
        x = patches.flatten(0, 1)
        for c in self.convs:
            x = c(x)
        x = x.view(b, t, *x.shape[1:])
        if aggregate:
            x = x.mean(1, keepdim=True).expand(-1, t, -1, -1, -1)
        return x.flatten(0, 1)

    def discriminate_features(self, f1, f2):
        x = torch.cat([f1.flatten(1), f2.flatten(1)], dim=1)
        for lin in self.pairlinear:
            x = lin(x)
        return x


def apply_random_crop(x, target_size, scale_range, num_crops=1):
    """util/util.py:323-343: per crop a random horizontal flip, independent x / y scale, an offset that keeps the window inside
    the image; bilinear grid_sample with zero padding, align_corners=False.  Random draws in the reference's order."""
    b = x.size(0) * num_crops
    # (the draws come from the CPU generator in fp32, as in the reference; `.to(x)` is a no-op there and lets a parity run
    # evaluate the restatement on another device / in double with the same crop windows)
    flip = (torch.round(torch.rand(b, 1, 1, 1)) * 2 - 1.0).to(x)
    gx = torch.linspace(-1.0, 1.0, target_size)[None, None, :, None].repeat(b, target_size, 1, 1).to(x)
    grid = torch.cat([gx * flip, gx.transpose(1, 2)], dim=3)
    x = x.unsqueeze(1).expand(-1, num_crops, -1, -1, -1).flatten(0, 1)
    scale = (torch.rand(b, 1, 1, 2) * (scale_range[1] - scale_range[0]) + scale_range[0]).to(x)
    offset = ((torch.rand(b, 1, 1, 2) * 2 - 1) * (1 - scale.cpu().float())).to(x)
    crop = F.grid_sample(x, grid * scale + offset, align_corners=False)
    return crop.view(b // num_crops, num_crops, *crop.shape[1:])


def gan_loss(pred, real):
    """models/networks/loss.py:10-16"""
    return F.softplus(-pred if real else pred).view(pred.size(0), -1).mean(dim=1)


class TrainIterationCPU:
    """The reference's model and driver for one discriminator call and one generator call (swapping_autoencoder_model.py:
    116-136,187-231; optimizers/swapping_autoencoder_optimizer.py:34-42,67-111 without the lazy-R1 call) on the CPU path."""

    def __init__(self, opt):
        self.opt = opt
        self.E, self.G = EncoderCPU(opt), GeneratorCPU(opt)
        self.D = DiscriminatorCPU(opt.crop_size, 2.0 * opt.netD_scale_capacity)
        self.Dpatch = PatchDiscriminatorCPU(opt)
        self.Gparams = list(self.G.parameters()) + list(self.E.parameters())
        self.Dparams = list(self.D.parameters()) + list(self.Dpatch.parameters())
        c = opt.R1_once_every / (1 + opt.R1_once_every)
        self.optimizer_G = torch.optim.Adam(self.Gparams, lr=opt.lr, betas=(opt.beta1, opt.beta2))
        self.optimizer_D = torch.optim.Adam(self.Dparams, lr=opt.lr * c, betas=(opt.beta1 ** c, opt.beta2 ** c))

    def modules(self):
        return {"E": self.E, "G": self.G, "D": self.D, "Dpatch": self.Dpatch}

    @staticmethod
    def swap(x):
        return torch.flip(x.view(x.shape[0] // 2, 2, *x.shape[1:]), [1]).view(x.shape)

    def crops(self, x):
        o = self.opt
        return apply_random_crop(x, o.patch_size, (o.patch_min_scale, o.patch_max_scale), num_crops=o.patch_num_crops)

    def _trainable(self, on, off):
        for p in on:
            p.requires_grad_(True)
        for p in off:
            p.requires_grad_(False)

    def discriminator_call(self, real):
        o = self.opt
        self._trainable(self.Dparams, self.Gparams)
        self.optimizer_D.zero_grad()
        sp, gl = self.E(real)
        b = real.size(0)
        rec = self.G(sp[:b // 2], gl[:b // 2])
        mix = self.G(self.swap(sp), gl)
        losses = {"D_real": gan_loss(self.D(real), True) * o.lambda_GAN,
                  "D_rec": gan_loss(self.D(rec), False) * (0.5 * o.lambda_GAN),
                  "D_mix": gan_loss(self.D(mix), False) * (0.5 * o.lambda_GAN)}
        real_feat = self.Dpatch.extract_features(self.crops(real), aggregate=o.patch_use_aggregation)
        target_feat = self.Dpatch.extract_features(self.crops(real))
        mix_feat = self.Dpatch.extract_features(self.crops(mix))
        losses["PatchD_real"] = gan_loss(self.Dpatch.discriminate_features(real_feat, target_feat), True) * o.lambda_PatchGAN
        losses["PatchD_mix"] = gan_loss(self.Dpatch.discriminate_features(real_feat, mix_feat), False) * o.lambda_PatchGAN
        sum(v.mean() for v in losses.values()).backward()
        self.optimizer_D.step()
        return {k: float(v.detach().mean()) for k, v in losses.items()}

    def generator_call(self, real):
        o = self.opt
        self._trainable(self.Gparams, self.Dparams)
        self.optimizer_G.zero_grad()
        b = real.size(0)
        sp, gl = self.E(real)
        rec = self.G(sp[:b // 2], gl[:b // 2])
        sp_mix = self.swap(sp)
        losses = {"G_L1": F.l1_loss(rec, real[:b // 2]) * o.lambda_L1}
        if o.crop_size >= 1024:
            real, gl, sp_mix = real[b // 2:], gl[b // 2:], sp_mix[b // 2:]
        mix = self.G(sp_mix, gl)
        losses["G_GAN_rec"] = gan_loss(self.D(rec), True) * (o.lambda_GAN * 0.5)
        losses["G_GAN_mix"] = gan_loss(self.D(mix), True) * (o.lambda_GAN * 1.0)
        real_feat = self.Dpatch.extract_features(self.crops(real), aggregate=o.patch_use_aggregation).detach()
        mix_feat = self.Dpatch.extract_features(self.crops(mix))
        losses["G_mix"] = gan_loss(self.Dpatch.discriminate_features(real_feat, mix_feat), True) * o.lambda_PatchGAN
        sum(v.mean() for v in losses.values()).backward()
        self.optimizer_G.step()
        return {k: float(v.detach().mean()) for k, v in losses.items()}
