"""StyleGAN2ResnetGenerator (reference: models/networks/generator.py:23-161, Fig. 18 of the paper).

Spatial code --(modulated by the global code)--> `netG_num_base_resnet_layers` resolution-
preserving styled residual blocks --> one upsampling styled residual block per encoder
downsampling (transposed 3x3 + blur main path, bilinear x2 skip) --> ToRGB.

Two MI355X-specific restructurings that leave parameters, checkpoint keys and results unchanged:

* every ModulatedConv2d (13 at the church preset) and the SpatialCodeModulation project the SAME
  global code through its own EqualLinear(global_code_ch -> C_in).  Launched one by one these are
  ~30 weight-bandwidth-bound [B x 2048] x [2048 x C] products per pass ("launch
  bound"); here their weights are concatenated per forward and ONE GEMM produces all style
  vectors (same products and sums per output element, so the values are unchanged);
* the skip path  F.interpolate(skip, x2, bilinear) -> (skip + res) / sqrt(2)  (:51-53) runs as
  one fused kernel (stylegan2_op.upsample2x_add).
"""
import math

import torch
import torch.nn.functional as F

from .. import util
from ..stylegan2_layers import ConvLayer, EqualLinear, ModulatedConv2d, StyledConv, ToRGB
from ..stylegan2_op import add_scale, linear, plane_affine, upsample2x_add
from ..stylegan2_op.modulate import ActTicket, GradScaleTicket
from .base_network import BaseNetwork

_INV_SQRT2 = 1.0 / math.sqrt(2)


class ResolutionPreservingResnetBlock(torch.nn.Module):
    """generator.py:23-36"""

    def __init__(self, opt, inch, outch, styledim):
        super().__init__()
        self.conv1 = StyledConv(inch, outch, 3, styledim, upsample=False)
        self.conv2 = StyledConv(outch, outch, 3, styledim, upsample=False)
        self.skip = ConvLayer(inch, outch, 1, activate=False, bias=False) if inch != outch else torch.nn.Identity()

    def forward(self, x, style):
        # conv1's activation has one consumer, conv2: its activation backward rides in conv2's backward (ActTicket)
        ticket = ActTicket()
        res = self.conv2(self.conv1(x, style, act_ticket=ticket), style, input_ticket=ticket)
        return add_scale(self.skip(x), res, _INV_SQRT2)


class UpsamplingResnetBlock(torch.nn.Module):
    """generator.py:39-53"""

    def __init__(self, inch, outch, styledim, blur_kernel=[1, 3, 3, 1], use_noise=False):
        super().__init__()
        self.inch, self.outch, self.styledim = inch, outch, styledim
        self.conv1 = StyledConv(inch, outch, 3, styledim, upsample=True, blur_kernel=blur_kernel, use_noise=use_noise)
        self.conv2 = StyledConv(outch, outch, 3, styledim, upsample=False, use_noise=use_noise)
        self.skip = ConvLayer(inch, outch, 1, activate=True, bias=True) if inch != outch else torch.nn.Identity()

    def forward(self, x, style):
        ticket = ActTicket()      # see ResolutionPreservingResnetBlock
        merge = GradScaleTicket()  # conv2's output has one consumer too, the merge: its 1/sqrt(2) rides in conv2's backward
        res = self.conv2(self.conv1(x, style, act_ticket=ticket), style, input_ticket=ticket, grad_scale_ticket=merge)
        # (bilinear_x2(skip) + res) / sqrt(2) in one pass
        return upsample2x_add(self.skip(x), res, _INV_SQRT2, res_ticket=merge)


class GeneratorModulation(torch.nn.Module):
    """x * scale(style) + bias(style) (generator.py:56-67)"""

    def __init__(self, styledim, outch):
        super().__init__()
        self.scale = EqualLinear(styledim, outch)
        self.bias = EqualLinear(styledim, outch)
        self._projected = None       # (scale(style), bias(style)) when the generator batched them

    def forward(self, x, style):
        if style.ndimension() <= 2:
            if self._projected is not None:
                sc, bi = self._projected
            else:
                sc, bi = self.scale(style), self.bias(style)
            return plane_affine(x, sc, bi)           # x * (1 * scale) + bias as one kernel
        style = F.interpolate(style, size=(x.size(2), x.size(3)), mode="bilinear", align_corners=False)
        return x * (1 * self.scale(style)) + self.bias(style)


class StyleGAN2ResnetGenerator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--netG_scale_capacity", default=1.0, type=float)
        parser.add_argument("--netG_num_base_resnet_layers", default=2, type=int)
        parser.add_argument("--netG_use_noise", type=util.str2bool, nargs="?", const=True, default=True)
        parser.add_argument("--netG_resnet_ch", type=int, default=256)
        return parser

    def __init__(self, opt):
        super().__init__(opt)
        blur_kernel = [1, 3, 3, 1] if opt.use_antialias else [1]
        self.global_code_ch = opt.global_code_ch + opt.num_classes
        self.add_module("SpatialCodeModulation", GeneratorModulation(self.global_code_ch, opt.spatial_code_ch))

        n_head = opt.netG_num_base_resnet_layers
        ch = opt.spatial_code_ch
        for i in range(n_head):   # widen gradually towards nf(0)
            out_ch = max(opt.spatial_code_ch, round((i + 1) / n_head * self.nf(0)))
            self.add_module("HeadResnetBlock%d" % i,
                            ResolutionPreservingResnetBlock(opt, ch, out_ch, self.global_code_ch))
            ch = out_ch

        for j in range(opt.netE_num_downsampling_sp):
            out_ch = self.nf(j + 1)
            self.add_module("UpsamplingResBlock%d" % (2 ** (4 + j)),
                            UpsamplingResnetBlock(ch, out_ch, self.global_code_ch, blur_kernel, opt.netG_use_noise))
            ch = out_ch

        self.add_module("ToRGB", ToRGB(ch, self.global_code_ch, blur_kernel=blur_kernel))

    def nf(self, num_up):
        """generator.py:141-144"""
        ch = 128 * (2 ** (self.opt.netE_num_downsampling_sp - num_up))
        return int(min(512, ch) * self.opt.netG_scale_capacity)

    # ---- batched style projections ----------------------------------------------------------------
    def _style_layers(self):
        mods = [m for m in self.modules() if isinstance(m, ModulatedConv2d)]
        lins = [self.SpatialCodeModulation.scale, self.SpatialCodeModulation.bias] + [m.modulation for m in mods]
        return mods, lins

    def _project_styles(self, global_code):
        """All EqualLinear(global_code) style projections of this pass as one GEMM."""
        mods, lins = self._style_layers()
        scale, lr_mul = lins[0].scale, lins[0].lr_mul
        if any(l.scale != scale or l.lr_mul != lr_mul or l.bias is None or l.activation for l in lins):
            return None, mods          # heterogeneous layers: let every module project for itself
        weight = torch.cat([l.weight for l in lins], dim=0)
        bias = torch.cat([l.bias for l in lins], dim=0) * lr_mul
        styles = linear(global_code, weight, bias=bias, alpha=scale)
        return torch.split(styles, [l.weight.shape[0] for l in lins], dim=1), mods

    def forward(self, spatial_code, global_code):
        spatial_code = util.normalize(spatial_code)
        global_code = util.normalize(global_code)
        pieces, mods = (None, [])
        if global_code.dim() == 2:
            pieces, mods = self._project_styles(global_code)
        try:
            if pieces is not None:
                self.SpatialCodeModulation._projected = (pieces[0], pieces[1])
                for m, s in zip(mods, pieces[2:]):
                    m._projected_style = s
            x = self.SpatialCodeModulation(spatial_code, global_code)
            for i in range(self.opt.netG_num_base_resnet_layers):
                x = getattr(self, "HeadResnetBlock%d" % i)(x, global_code)
            for j in range(self.opt.netE_num_downsampling_sp):
                x = getattr(self, "UpsamplingResBlock%d" % (2 ** (4 + j)))(x, global_code)
            return self.ToRGB(x, global_code, None)
        finally:
            self.SpatialCodeModulation._projected = None
            for m in mods:
                m._projected_style = None
