"""The lazy-R1 call in isolation: kernel time by kernel name for `compute_R1_loss` + backward
(torch.profiler device time, all kernels).  python tools/r1_profile.py > gpurun_out/r1_profile.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd.options import make_options  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model  # noqa: E402

opt = make_options("church256", batch_size=16, num_gpus=1)
torch.manual_seed(0)
model = create_model(opt)
net = model.singlegpu_model if hasattr(model, "singlegpu_model") else model
x = torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1
params = [p for n, p in net.named_parameters() if n.startswith("D.") or n.startswith("Dpatch.")]


def r1_step():
    for p in net.parameters():
        p.grad = None
    losses = model(x, command="compute_R1_loss")
    loss = sum(v.mean() for v in losses.values()) * 16
    loss.backward()


for _ in range(2):
    r1_step()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); r1_step(); t1.record(); torch.cuda.synchronize()
print("R1 step: %.1f ms" % t0.elapsed_time(t1))
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    r1_step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("device time total %.1f ms" % (tot / 1e3))
for e in rows[:40]:
    print("%9.1f us  n=%-4d %s" % (e.self_device_time_total, e.count, e.key[:110]))
