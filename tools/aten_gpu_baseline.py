"""Stock PyTorch-ROCm (ATen / MIOpen) on the same GPU as a comparison point ("the reference modules on-device
through ATen/MIOpen"): the image discriminator forward + backward of oracle/aten_cpu_path.py (the reference's code path with
the custom-kernel gate off, pinned to the reference's own Discriminator) on cuda:0, next to the same pass on this repo's
kernels.  Baseline infrastructure, like bench.py's cpu_baseline leg.   python tools/aten_gpu_baseline.py [batch]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aten_cpu_path as A  # noqa: E402
from swapping_autoencoder_pytorch_amd.stylegan2_layers import Discriminator  # noqa: E402
from swapping_autoencoder_pytorch_amd import loss  # noqa: E402


def timed(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.rand(batch, 3, 256, 256, device=dev) * 2 - 1
    stock = A.DiscriminatorCPU(256, 2).to(dev)
    ours = Discriminator(256, 2).to(dev)
    flops = stock.train_flops(batch, 256)

    def run(net, lossfn):
        def f():
            for p in net.parameters():
                p.grad = None
            lossfn(net(x)).mean().backward()
        return f

    t_stock = timed(run(stock, lambda p: torch.nn.functional.softplus(-p)))
    t_ours = timed(run(ours, lambda p: loss.gan_loss(p, True)))
    print(json.dumps({"what": "image discriminator forward + backward, %d x 3 x 256 x 256, fp32, weights trainable" % batch,
                      "conv_tflop": round(flops / 1e12, 3),
                      "stock_pytorch_rocm_ms": round(t_stock * 1e3, 2), "stock_tflops": round(flops / t_stock / 1e12, 2),
                      "this_repo_ms": round(t_ours * 1e3, 2), "this_repo_tflops": round(flops / t_ours / 1e12, 2),
                      "speedup": round(t_stock / t_ours, 2), "torch": torch.__version__}))


if __name__ == "__main__":
    main()
