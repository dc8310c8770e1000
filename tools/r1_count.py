"""Count conv launches by (op, geometry) in one lazy-R1 call (debug aid)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd.options import make_options  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model  # noqa: E402
from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as cg  # noqa: E402

opt = make_options("church256", batch_size=16, num_gpus=1)
torch.manual_seed(0)
model = create_model(opt)
net = model.singlegpu_model if hasattr(model, "singlegpu_model") else model
x = torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1
counts = collections.Counter()
phase = ["fwd+grad"]
orig = cg._launch


def launch(name, op, geom, a, b, out_shape):
    counts[(phase[0], ("fwd", "dgrad", "wgrad")[op], geom.n, geom.c, geom.h, geom.m, geom.k, geom.stride)] += 1
    return orig(name, op, geom, a, b, out_shape)


cg._launch = launch
orig_f = cg._launch_fused


def fused(geom, *a):
    counts[(phase[0], "fwd+act", geom.n, geom.c, geom.h, geom.m, geom.k, geom.stride)] += 1
    return orig_f(geom, *a)


cg._launch_fused = fused
losses = model(x, command="compute_R1_loss")
loss = sum(v.mean() for v in losses.values()) * 16
phase[0] = "backward"
loss.backward()
torch.cuda.synchronize()
for k, v in sorted(counts.items()):
    if k[6] == 3 and k[7] == 1:
        print(v, k)
