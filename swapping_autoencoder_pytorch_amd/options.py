"""Option namespace of the hot path: the flags (and defaults) the reference's staged argparse
contributes for the training step (options/__init__.py:18-53 base flags, plus the
modify_commandline_options of the model, the four networks and the optimizer), without the
dataset / visualizer / evaluator / launcher flags that are out of scope.

``make_options(**overrides)`` returns an argparse.Namespace usable wherever the reference passes
``opt``.  Presets mirror the reference launchers that BASELINE.json's configs name."""
import argparse

from . import util
from .networks import find_network_using_name


def build_parser():
    p = argparse.ArgumentParser(add_help=False)
    # options/__init__.py:18-53
    p.add_argument("--name", type=str, default="mi355x")
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--checkpoints_dir", type=str, default="./checkpoints/")
    p.add_argument("--model", type=str, default="swapping_autoencoder")
    p.add_argument("--optimizer", type=str, default="swapping_autoencoder")
    p.add_argument("--phase", type=str, default="train")
    p.add_argument("--resume_iter", type=str, default="latest")
    p.add_argument("--num_classes", type=int, default=0)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--load_size", type=int, default=256)
    p.add_argument("--crop_size", type=int, default=256)
    p.add_argument("--netG", default="StyleGAN2Resnet")
    p.add_argument("--netD", default="StyleGAN2")
    p.add_argument("--netE", default="StyleGAN2Resnet")
    p.add_argument("--netPatchD", default="StyleGAN2")
    p.add_argument("--use_antialias", type=util.str2bool, default=True)
    # TrainOptions (options/__init__.py:150-165)
    p.add_argument("--continue_train", type=util.str2bool, default=False)
    p.add_argument("--pretrained_name", type=str, default=None)
    # model, networks, optimizer
    from .swapping_autoencoder_model import SwappingAutoencoderModel
    from .swapping_autoencoder_optimizer import SwappingAutoencoderOptimizer
    SwappingAutoencoderModel.modify_commandline_options(p, True)
    for name, mode in (("StyleGAN2Resnet", "encoder"), ("StyleGAN2Resnet", "generator"), ("StyleGAN2", "discriminator"),
                       ("StyleGAN2", "patch_discriminator")):
        find_network_using_name(name, mode).modify_commandline_options(p, True)
    SwappingAutoencoderOptimizer.modify_commandline_options(p, True)
    return p


# reference launcher presets (experiments/*_launcher.py) restricted to hot-path flags
PRESETS = {
    # church_launcher.py:5-23 (bedroom_launcher.py uses the same networks)
    "church256": dict(crop_size=256, load_size=256, patch_use_aggregation=False),
    "bedroom256": dict(crop_size=256, load_size=256),
    # ffhq_launcher.py:12-22
    "ffhq512": dict(crop_size=512, load_size=512, netG_scale_capacity=1.0, netE_num_downsampling_sp=4,
                    netE_num_downsampling_gl=2, lambda_patch_R1=10.0),
    # ffhq_launcher.py:24-34 at its 1024 test resolution
    "ffhq1024": dict(crop_size=1024, load_size=1024, netG_scale_capacity=0.8, netE_num_downsampling_sp=5,
                     netE_scale_capacity=0.4, global_code_ch=1536, patch_size=256, lambda_patch_R1=10.0),
    # BASELINE.json config 1 (sp-downsampling 2, patch 32)
    "tiny32": dict(crop_size=32, load_size=32, netE_num_downsampling_sp=2, patch_size=32),
}


def make_options(preset=None, **overrides):
    opt = build_parser().parse_args([])
    if preset is not None:
        for k, v in PRESETS[preset].items():
            setattr(opt, k, v)
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError("unknown option %r" % k)
        setattr(opt, k, v)
    opt.isTrain = True
    return opt
