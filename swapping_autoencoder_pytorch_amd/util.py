"""The six helpers of the reference's util/util.py that sit on the training hot path
(SURVEY.md §2 row 18), restated: normalize (:18-22), apply_random_crop (:323-343), to_numpy
(:423-429), str2bool (:43-51).  ``is_custom_kernel_supported`` (:432-436) is deliberately absent:
the HIP kernels are the only implementation, there is nothing to gate."""
import argparse

import torch

from . import rng
from .stylegan2_op import l2_normalize, random_crop


def normalize(v):
    """L2-normalise over dim 1 (epsilon inside the rsqrt, as the reference)."""
    if isinstance(v, list):
        return [normalize(x) for x in v]
    return l2_normalize(v, 1e-8)        # v * rsqrt(sum(v ** 2, dim=1, keepdim=True) + 1e-8) as one kernel


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def apply_random_crop(x, target_size, scale_range, num_crops=1):
    """``num_crops`` random crops per image, each with a random horizontal flip, independent x/y
    scale in ``scale_range`` and a random offset that keeps the window inside the image, resampled
    bilinearly to ``target_size`` (grid_sample, align_corners=False, zero padding).

    Consumes the RNG exactly as the reference does (flip [B,1,1,1], scale [B,1,1,2], offset
    [B,1,1,2] — SURVEY.md Appendix B) so identical seeds give identical crops."""
    b = x.size(0) * num_crops
    dev = x.device
    flip = torch.round(rng.rand((b, 1, 1, 1), dev)) * 2 - 1.0
    # same RNG draws in the same order as the reference, then the sampler kernel (stylegan2_op.random_crop)
    # instead of expand + grid + F.grid_sample.  First-order differentiable, which is all the train step asks of
    # it (the R1 crops are detached leaves, swapping_autoencoder_model.py:206-207).  There is no ATen path: a
    # non-fp32 or non-GPU image is refused by the library binding like everywhere else in the product.
    scale = rng.rand((b, 1, 1, 2), dev) * (scale_range[1] - scale_range[0]) + scale_range[0]
    offset = (rng.rand((b, 1, 1, 2), dev) * 2 - 1) * (1 - scale)
    crop = random_crop(x, flip, scale, offset, target_size, num_crops)
    return crop.view(b // num_crops, num_crops, crop.size(1), crop.size(2), crop.size(3))


def to_numpy(metric_dict):
    out = {}
    for k, v in metric_dict.items():
        if "numpy" not in str(type(v)):
            v = v.detach().cpu().mean().numpy()
        out[k] = v
    return out
