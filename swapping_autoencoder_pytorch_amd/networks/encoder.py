"""StyleGAN2ResnetEncoder (reference: models/networks/encoder.py:31-114).

FromRGB 1x1 -> `netE_num_downsampling_sp` reflection-padded ResBlocks with the light [1,2,1]
blur -> two 1x1 convs to the spatial (structure) code; global (texture) branch: stride-2 3x3
ConvLayers without blur -> spatial mean -> EqualLinear.  Both codes are L2-normalised."""
import torch.nn as nn
import torch.nn.functional as F

from .. import util
from ..stylegan2_layers import ConvLayer, EqualLinear, ResBlock
from .base_network import BaseNetwork


class StyleGAN2ResnetEncoder(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--netE_scale_capacity", default=1.0, type=float)
        parser.add_argument("--netE_num_downsampling_sp", default=4, type=int)
        parser.add_argument("--netE_num_downsampling_gl", default=2, type=int)
        parser.add_argument("--netE_nc_steepness", default=2.0, type=float)
        return parser

    def __init__(self, opt):
        super().__init__(opt)
        n_sp, n_gl = opt.netE_num_downsampling_sp, opt.netE_num_downsampling_gl
        blur_kernel = [1, 2, 1] if opt.use_antialias else [1]

        self.add_module("FromRGB", ConvLayer(3, self.nc(0), 1))

        self.DownToSpatialCode = nn.Sequential()
        for i in range(n_sp):
            self.DownToSpatialCode.add_module(
                "ResBlockDownBy%d" % (2 ** i), ResBlock(self.nc(i), self.nc(i + 1), blur_kernel, reflection_pad=True))

        ch = self.nc(n_sp)
        self.add_module("ToSpatialCode", nn.Sequential(
            ConvLayer(ch, ch, 1, activate=True, bias=True),
            ConvLayer(ch, opt.spatial_code_ch, kernel_size=1, activate=False, bias=True)))

        self.DownToGlobalCode = nn.Sequential()
        for i in range(n_gl):
            j = n_sp + i
            self.DownToGlobalCode.add_module(
                "ConvLayerDownBy%d" % (2 ** j),
                ConvLayer(self.nc(j), self.nc(j + 1), kernel_size=3, blur_kernel=[1], downsample=True, pad=0))

        self.add_module("ToGlobalCode", nn.Sequential(EqualLinear(self.nc(n_sp + n_gl), opt.global_code_ch)))

    def nc(self, idx):
        """Channel width after `idx` downsamplings (encoder.py:87-91)."""
        nc = self.opt.netE_nc_steepness ** (5 + idx) * self.opt.netE_scale_capacity
        return round(min(self.opt.global_code_ch, int(round(nc))))

    def forward(self, x, extract_features=False):
        midpoint = self.DownToSpatialCode(self.FromRGB(x))
        sp = self.ToSpatialCode(midpoint)

        if extract_features:
            padded = F.pad(midpoint, (1, 0, 1, 0), mode="reflect")
            feature = self.DownToGlobalCode[0](padded)
            assert feature.size(2) == sp.size(2) // 2 and feature.size(3) == sp.size(3) // 2
            feature = F.interpolate(feature, size=(7, 7), mode="bilinear", align_corners=False)

        gl = self.ToGlobalCode(self.DownToGlobalCode(midpoint).mean(dim=(2, 3)))
        sp, gl = util.normalize(sp), util.normalize(gl)
        return (sp, gl, feature) if extract_features else (sp, gl)
