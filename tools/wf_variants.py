"""Same-box timing of csrc/winograd_fused.hip variants (tools/build_wf_variant.sh): the one-kernel Winograd convolution and weight
gradient on three of the step's shapes per library.   python tools/wf_variants.py product wf_prev ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402
from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as cg, winograd  # noqa: E402

SHAPES = [(16, 512, 64), (16, 256, 128), (16, 128, 256), (128, 128, 32), (40, 128, 256), (8, 256, 128), (40, 512, 64), (128, 64, 64)]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    for rnd in range(2):
        for name in sys.argv[1:]:
            path = hip_lib.DEFAULT_LIBRARY if name == "product" else os.path.join(ROOT, "tools", "variants", name + ".so")
            hip_lib._LIB = hip_lib.SaeLibrary(path)
            out = []
            for n, c, side in SHAPES:
                g = cg._Geom(n, c, side, side, c, 3, 1, 1, False, 1.0 / (c * 9) ** 0.5)
                x = torch.randn(n, c, side, side, device="cuda")
                gy = torch.randn(n, c, side, side, device="cuda")
                w = torch.nn.Parameter(torch.randn(c, c, 3, 3, device="cuda"))
                ex = 32.0 * n * c * c * (side // 2) ** 2
                tf = timed(lambda: winograd.conv(x, w, g, kind="fused"))
                tw = timed(lambda: winograd.wgrad(x, gy, g, kind="fused"))
                out.append("%dch@%d n%d: conv %.3f ms (%.3f) wgrad %.3f ms (%.3f)" % (c, side, n, tf, ex / tf / 1e9 / 157.3, tw, ex / tw / 1e9 / 157.3))
            print("%-12s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
