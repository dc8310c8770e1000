"""Train the same model from the same seed on the same synthetic batches under both conv arithmetics and report
how far the loss trajectories drift apart (evidence for DESIGN.md section 4, bf16x6).
python tools/compare_math.py [preset] [batch] [iters]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402
from swapping_autoencoder_pytorch_amd.options import make_options  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "church256"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6


def run(mode):
    hip_lib.set_conv_math(mode)
    opt = make_options(preset, batch_size=batch, num_gpus=1)
    torch.manual_seed(0)
    model = create_model(opt)
    optimizer = create_optimizer(opt, model)
    torch.manual_seed(1)
    g = torch.Generator(device="cuda").manual_seed(2)
    hist = []
    for i in range(iters):
        for _ in range(2):          # D call, G call
            x = torch.rand(batch, 3, opt.crop_size, opt.crop_size, device="cuda", generator=g) * 2 - 1
            losses = optimizer.train_one_step({"real_A": x}, i)
            hist.append({k: float(v) for k, v in losses.items()})
    hip_lib.set_conv_math("f32")
    return hist


a = run("f32")
b = run("f32")          # run-to-run repeatability of the exact arithmetic (same seeds)
c = run("bf16x6")
for name, other in (("f32 vs f32 (repeat)", b), ("f32 vs bf16x6", c)):
    worst = []
    for step, (u, v) in enumerate(zip(a, other)):
        d = max(abs(u[k] - v[k]) / max(abs(u[k]), 1e-3) for k in u if k in v)
        worst.append(d)
    print(json.dumps({"compare": name, "preset": preset, "batch": batch,
                      "max_rel_loss_diff_per_call": [float("%.3g" % w) for w in worst]}))
