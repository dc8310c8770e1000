"""A/B timing of library variants on the K2 backward (bias_act_bwd) and K1 blur shapes of the church256 step:
   python tools/ab_k2.py base new ...   (names under tools/variants/, or "product")"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd import hip_lib as L  # noqa: E402

dev = torch.device("cuda:0")
K2 = [(40, 128, 256, 256), (24, 128, 256, 256), (384, 32, 128, 128), (40, 256, 128, 128), (40, 512, 64, 64), (384, 64, 64, 64),
      (40, 512, 32, 32), (384, 128, 32, 32)]
K1 = [(40 * 128, 256, 256, 4, 2), (40 * 128, 256, 256, 4, 1), (16 * 128, 257, 257, 4, 1), (384 * 32, 128, 128, 4, 2),
      (40 * 256, 128, 128, 4, 2), (40 * 512, 64, 64, 4, 2)]


def libpath(name):
    return L.DEFAULT_LIBRARY if name == "product" else os.path.join(ROOT, "tools", "variants", name + ".so")


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    names = sys.argv[1:] or ["product"]
    libs = [L.SaeLibrary(libpath(n)) for n in names]
    st = lambda: torch.cuda.current_stream(dev).cuda_stream
    print("%-34s " % "shape" + " ".join("%9s" % n for n in names) + "   (TB/s algorithmic; * = differs from the first)")
    for shape in K2:
        g = torch.randn(*shape, device=dev); y = torch.randn(*shape, device=dev)
        step = shape[2] * shape[3]
        best, outs = [1e9] * len(libs), []
        for rnd in range(2):
            for i, lib in enumerate(libs):
                gx = torch.empty_like(g); gb = torch.empty(shape[1], device=dev)
                n = lib.query("bias_act_bwd_workspace", g.numel(), step, shape[1])
                ws = torch.empty(max(n, 1), device=dev)
                fn = lambda: lib.call("bias_act_bwd_f32", g.data_ptr(), y.data_ptr(), gx.data_ptr(), gb.data_ptr(), ws.data_ptr(), n,
                                      g.numel(), step, shape[1], 0.2, 2 ** 0.5, st())
                best[i] = min(best[i], timeit(fn))
                if rnd == 0:
                    outs.append((gx, gb))
        same = [bool(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])) for o in outs]
        print("%-34s " % ("bias_act_bwd %s" % (shape,)) + " ".join("%8.2f%s" % (12.0 * g.numel() / t / 1e9, " " if ok else "*")
                                                                     for t, ok in zip(best, same)), flush=True)
    for (planes, h, w, k, pad) in K1:
        x = torch.randn(planes, h, w, device=dev)
        kk = torch.ones(k, k, device=dev) / (k * k)
        oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
        best, outs = [1e9] * len(libs), []
        for rnd in range(2):
            for i, lib in enumerate(libs):
                yv = torch.empty(planes, oh, ow, device=dev)
                fn = lambda: lib.call("upfirdn2d_f32", x.data_ptr(), kk.data_ptr(), yv.data_ptr(), planes, h, w, 1, k, k, 1, 1, 1, 1,
                                      pad, pad, pad, pad, st())
                best[i] = min(best[i], timeit(fn))
                if rnd == 0:
                    outs.append(yv)
        same = [bool(torch.equal(o, outs[0])) for o in outs]
        print("%-34s " % ("blur %dx%dx%d k%d pad%d" % (planes, h, w, k, pad)) +
              " ".join("%8.2f%s" % (4.0 * (x.numel() + outs[0].numel()) / t / 1e9, " " if ok else "*") for t, ok in zip(best, same)), flush=True)


if __name__ == "__main__":
    main()
