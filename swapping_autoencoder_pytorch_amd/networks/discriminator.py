"""StyleGAN2Discriminator: thin wrapper around the layer library's residual discriminator
(reference: models/networks/discriminator.py:5-31)."""
from ..stylegan2_layers import Discriminator as _ResidualDiscriminator
from .base_network import BaseNetwork


class StyleGAN2Discriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--netD_scale_capacity", default=1.0, type=float)
        return parser

    def __init__(self, opt):
        super().__init__(opt)
        self.stylegan2_D = _ResidualDiscriminator(
            opt.crop_size, 2.0 * opt.netD_scale_capacity,
            blur_kernel=[1, 3, 3, 1] if opt.use_antialias else [1])

    def forward(self, x):
        return self.stylegan2_D(x)

    def get_features(self, x):
        return self.stylegan2_D.get_features(x)

    def get_pred_from_features(self, feat, label):
        assert label is None
        return self.stylegan2_D.final_linear(feat.flatten(1))
