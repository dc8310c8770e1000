"""One D+G iteration under torch.profiler: device time of ATen ops grouped by op and input shapes
(to find the unfused elementwise glue).  python tools/torch_profile.py > gpurun_out/torch_profile.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd.options import make_options  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer  # noqa: E402

opt = make_options("church256", batch_size=16, num_gpus=1)
torch.manual_seed(0)
model = create_model(opt)
optimizer = create_optimizer(opt, model)
x = torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1
for i in range(2):
    optimizer.train_one_step({"real_A": x}, i)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    optimizer.train_one_step({"real_A": x}, 2)
    optimizer.train_one_step({"real_A": x}, 3)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("aten self device time total: %.2f ms for 2 train_one_step calls (one D+G iteration)" % (tot / 1e3))
for e in rows[:60]:
    print("%-34s %9.1f us  n=%-4d %s" % (e.key, e.self_device_time_total, e.count, str(e.input_shapes)[:150]))
