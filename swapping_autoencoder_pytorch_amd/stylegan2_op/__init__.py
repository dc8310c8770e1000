"""MI355X replacement of the reference's ``models.networks.stylegan2_op`` package (same public
names, models/networks/stylegan2_op/__init__.py:1-2) plus the dense conv / linear operators the
layer library is built on."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import blur_noise_bias_act, upfirdn2d
from .conv2d_gemm import (conv2d, conv2d_bias_act, conv_transpose2d, input_grads_only, linear, modulated_conv2d,
                          styled_modulated_conv2d)
from .upsample import add_scale, upsample2x_add
from .modulate import fusable, noise_bias_act, plane_scale
from .crop import random_crop
from .glue import l2_normalize, plane_affine, softplus_mean
from .pad import ReflectionPad2d, reflect_pad

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d", "blur_noise_bias_act", "conv2d", "conv2d_bias_act", "conv_transpose2d", "input_grads_only", "linear", "modulated_conv2d", "styled_modulated_conv2d", "upsample2x_add", "add_scale", "noise_bias_act", "plane_scale", "fusable", "random_crop", "reflect_pad", "ReflectionPad2d", "l2_normalize", "plane_affine", "softplus_mean"]
