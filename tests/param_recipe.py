"""Deterministic, constructor-order-independent parameter fill shared by the golden generator
(run on the REFERENCE modules) and the parity tests (run on OUR modules): every tensor of a
state_dict is drawn from a generator seeded by the CRC of its key, so identical keys + shapes
give identical values on both sides without shipping weights."""
import zlib

import torch

MICRO = dict(crop_size=32, load_size=32, batch_size=4, netE_num_downsampling_sp=2, patch_size=32,
             patch_num_crops=2, global_code_ch=64, netE_scale_capacity=0.25, netG_scale_capacity=0.125,
             netD_scale_capacity=0.03125, netPatchD_scale_capacity=0.5, netPatchD_max_nc=32, num_gpus=0,
             R1_once_every=2)


def fill_params(module, seed=0):
    with torch.no_grad():
        for key, t in module.state_dict().items():
            if key.endswith("kernel") or key.endswith("num_discriminator_iters") or not t.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
            if t.dim() >= 2 and key.endswith("weight"):
                v = torch.randn(t.shape, generator=g)
            elif key.endswith("modulation.bias"):
                v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            else:
                v = 0.1 * torch.randn(t.shape, generator=g)
            t.copy_(v.to(t.device))


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def uniform_images(b, size, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, 3, size, size, generator=g) * 2 - 1
