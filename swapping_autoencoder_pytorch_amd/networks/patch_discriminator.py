"""StyleGAN2PatchDiscriminator (reference: models/networks/patch_discriminator.py:97-171): a
residual feature extractor over 128x128 crops down to a [C, 2, 2] descriptor, and a 4-layer MLP
that scores a (reference patch, query patch) descriptor pair.  The dead code paths of the
reference's BasePatchDiscriminator (:35-93) are not reproduced."""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import util
from ..stylegan2_layers import ConvLayer, EqualLinear, ResBlock, run_sequence
from .base_network import BaseNetwork


class BasePatchDiscriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--netPatchD_scale_capacity", default=4.0, type=float)
        parser.add_argument("--netPatchD_max_nc", default=256 + 128, type=int)
        parser.add_argument("--patch_size", default=128, type=int)
        parser.add_argument("--max_num_tiles", default=8, type=int)
        parser.add_argument("--patch_random_transformation", type=util.str2bool, nargs="?", const=True, default=False)
        return parser

    def needs_regularization(self):
        return False

    def extract_features(self, patches):
        raise NotImplementedError()

    def discriminate_features(self, feature1, feature2):
        raise NotImplementedError()


class StyleGAN2PatchDiscriminator(BasePatchDiscriminator):
    def __init__(self, opt):
        super().__init__(opt)
        cm, max_nc, size = opt.netPatchD_scale_capacity, opt.netPatchD_max_nc, opt.patch_size
        channels = {4: min(max_nc, int(256 * cm)), 8: min(max_nc, int(128 * cm)), 16: min(max_nc, int(64 * cm)),
                    32: int(32 * cm), 64: int(16 * cm), 128: int(8 * cm), 256: int(4 * cm)}
        log_size = int(math.ceil(math.log(size, 2)))
        blur_kernel = [1, 3, 3, 1] if opt.use_antialias else [1]

        ch = channels[2 ** log_size]
        convs = [("0", ConvLayer(3, ch, 3))]
        for i in range(log_size, 2, -1):
            out_ch = channels[2 ** (i - 1)]
            name = str(7 - i) if i <= 6 else "%dx%d" % (2 ** i, 2 ** i)
            convs.append((name, ResBlock(ch, out_ch, blur_kernel)))
            ch = out_ch
        convs.append(("5", ResBlock(ch, max_nc * 2, downsample=False)))
        convs.append(("6", ConvLayer(max_nc * 2, max_nc, 3, pad=0)))
        self.convs = nn.Sequential(OrderedDict(convs))

        self.pairlinear = nn.Sequential(
            EqualLinear(channels[4] * 2 * 2 * 2, 2048, activation="fused_lrelu"),
            EqualLinear(2048, 2048, activation="fused_lrelu"),
            EqualLinear(2048, 1024, activation="fused_lrelu"),
            EqualLinear(1024, 1))

    def extract_features(self, patches, aggregate=False):
        if patches.ndim == 5:
            b, t = patches.shape[:2]
            flat = patches.flatten(0, 1)
        else:
            b, t = patches.size(0), patches.size(1)
            flat = patches
        feats = run_sequence(self.convs, flat)
        feats = feats.view(b, t, *feats.shape[1:])
        if aggregate:
            feats = feats.mean(1, keepdim=True).expand(-1, t, -1, -1, -1)
        return feats.flatten(0, 1)

    def extract_layerwise_features(self, image):
        feats = [image]
        for m in self.convs:
            feats.append(m(feats[-1]))
        return feats

    def discriminate_features(self, feature1, feature2):
        return self.pairlinear(torch.cat([feature1.flatten(1), feature2.flatten(1)], dim=1))
