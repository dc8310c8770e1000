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
                bias = torch.randn(c, device="cuda")
                xs = torch.rand(n, c, device="cuda") + 0.5
                os_ = torch.rand(n, c, device="cuda") + 0.5
                nz = torch.randn(n, 1, side, side, device="cuda")
                nw = torch.full((1,), 0.1, device="cuda")
                tf = timed(lambda: winograd.conv(x, w, g, kind="fused"))
                ta = timed(lambda: winograd.conv(x, w, g, bias=bias, act=(0.2, 2 ** 0.5), kind="fused"))
                tm = timed(lambda: winograd.conv(x, w, g, bias=bias, act=(0.2, 2 ** 0.5), x_scale=xs, out_scale=os_, noise=nz,
                                                 noise_weight=nw, kind="fused"))
                tw = timed(lambda: winograd.wgrad(x, gy, g, kind="fused"))
                fr = lambda t: ex / t / 1e9 / 157.3
                out.append("%dch@%d n%d: conv %.3f ms (%.3f) +bias+act %.3f (%.3f) modulated+noise %.3f (%.3f) wgrad %.3f ms (%.3f)"
                           % (c, side, n, tf, fr(tf), ta, fr(ta), tm, fr(tm), tw, fr(tw)))
            print("%-12s %s" % (name, "\n             ".join(out)), flush=True)


if __name__ == "__main__":
    main()
