"""Host-side profile of the tiny32 step (pure launch overhead: 32x32 images, B=4): cProfile over N iterations, top functions by
internal time.   python tools/host_profile.py [iterations]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd.options import make_options  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    opt = make_options("tiny32", batch_size=4, num_gpus=1)
    torch.manual_seed(0)
    model = create_model(opt)
    optimizer = create_optimizer(opt, model)
    pool = [torch.rand(4, 3, 32, 32, device="cuda") * 2 - 1 for _ in range(4)]

    def it(i):
        optimizer.train_one_step({"real_A": pool[(2 * i) % 4]}, i)
        optimizer.train_one_step({"real_A": pool[(2 * i + 1) % 4]}, i)
    for i in range(5):
        it(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, 5 + iters):
        it(i)
    torch.cuda.synchronize()
    print("tiny32: %.2f ms per D+G iteration without the profiler" % ((time.perf_counter() - t0) / iters * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(5 + iters, 5 + 2 * iters):
        it(i)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
