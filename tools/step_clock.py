"""Shader clock of the fp32 MFMA kernels INSIDE the train step (not in an isolated loop): the church256 step is run on the
profiling variant of the library (tools/build_variant.sh clk ... -DSAE_CLOCK_PROBE), whose gather / wgrad workgroups
accumulate s_memtime (shader cycles) and s_memrealtime (100 MHz) -- average MHz over all their workgroups of N iterations,
plus the wave-0 phase split summed over those kernels.   python tools/step_clock.py [iterations]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402

CLK = os.path.join(ROOT, "tools", "variants", "clk.so")
hip_lib._LIB = hip_lib.SaeLibrary(CLK)          # every op of this process goes through the instrumented build
from swapping_autoencoder_pytorch_amd.options import make_options  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model  # noqa: E402
from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda", 0)
    raw = C.CDLL(CLK)
    raw.sae_debug_clock_probe.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 10)()
    opt = make_options("church256", batch_size=16, num_gpus=1)
    torch.manual_seed(0)
    model = create_model(opt)
    optimizer = create_optimizer(opt, model)
    pool = [torch.rand(16, 3, 256, 256, device=dev) * 2 - 1 for _ in range(4)]
    for i in range(3):
        optimizer.train_one_step({"real_A": pool[(2 * i) % 4]}, i)
        optimizer.train_one_step({"real_A": pool[(2 * i + 1) % 4]}, i)
    torch.cuda.synchronize()
    raw.sae_debug_clock_probe(buf, 1)
    t0 = time.perf_counter()
    for i in range(3, 3 + iters):
        optimizer.train_one_step({"real_A": pool[(2 * i) % 4]}, i)
        optimizer.train_one_step({"real_A": pool[(2 * i + 1) % 4]}, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    raw.sae_debug_clock_probe(buf, 1)
    mhz = buf[0] / (buf[1] / 100e6) / 1e6
    ph = [buf[3 + i] for i in range(7)]
    tot = float(sum(ph)) or 1.0
    print("church256 B=16, %d iterations at %.1f ms (instrumented build): %d workgroups of the fp32 MFMA kernels ran at %.0f MHz on "
          "average (2400 = the clock the 157.3 TFLOP/s peak assumes)" % (iters, dt * 1e3, buf[2], mhz))
    print("wave-0 time of those workgroups: prologue %.1f | MFMA %.1f | barrier %.1f | LDS stores %.1f | barrier %.1f | load issue %.1f | "
          "epilogue %.1f  (%%)" % tuple(100.0 * v / tot for v in ph))


if __name__ == "__main__":
    main()
