"""Same-box A/B of library variants on the K1 launches with a fused epilogue (sae_upfirdn2d_epilogue_f32: accumulate, activation
backward + bias-gradient partials; sae_upfirdn2d_noise_bias_act_f32) on the step's big planes.  TB/s of algorithmic bytes
(4 x (numel_in + numel_out [+ old y] [+ act_ref])).    python tools/ab_k1_epilogue.py product k1_head ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd import hip_lib as L  # noqa: E402

dev = torch.device("cuda:0")
# major, in_h, in_w, k, up, pad0, pad1, channels, accumulate, act_ref
EPI = [
    (5120, 128, 128, 4, 2, 2, 1, 128, 1, 1, "x2 + accumulate + act bwd 128^2 n40"),
    (3072, 128, 128, 4, 2, 2, 1, 128, 1, 1, "x2 + accumulate + act bwd 128^2 n24"),
    (12288, 64, 64, 4, 2, 2, 1, 32, 1, 1, "x2 + accumulate + act bwd 64^2 Dp"),
    (5120, 257, 257, 4, 1, 1, 1, 128, 0, 1, "blur + act bwd 257^2 n40"),
    (12288, 129, 129, 4, 1, 1, 1, 32, 0, 1, "blur + act bwd 129^2 Dp"),
    (10240, 129, 129, 4, 1, 1, 1, 256, 0, 1, "blur + act bwd 129^2 n40"),
    (24576, 65, 65, 4, 1, 1, 1, 64, 0, 1, "blur + act bwd 65^2 Dp"),
    (10240, 64, 64, 4, 2, 2, 1, 256, 1, 0, "x2 + accumulate 64^2 n40"),
    (24576, 32, 32, 4, 2, 2, 1, 64, 1, 0, "x2 + accumulate 32^2 Dp"),
    (20480, 32, 32, 4, 2, 2, 1, 512, 1, 0, "x2 + accumulate 32^2 n40"),
    (49152, 16, 16, 4, 2, 2, 1, 128, 1, 0, "x2 + accumulate 16^2 Dp"),
]
FWD = [(2048, 257, 257, 128, "blur + noise + bias + lrelu fwd 257^2"), (4096, 129, 129, 256, "blur + noise + bias + lrelu fwd 129^2"),
       (8192, 65, 65, 512, "blur + noise + bias + lrelu fwd 65^2")]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    names = sys.argv[1:] or ["product"]
    libs = [L.SaeLibrary(L.DEFAULT_LIBRARY if n == "product" else os.path.join(ROOT, "tools", "variants", n + ".so")) for n in names]
    st = torch.cuda.current_stream(dev).cuda_stream
    print("%-40s " % "launch" + " ".join("%10s" % n for n in names) + "   (TB/s; * = result differs from the first)")
    for major, h, w, k, up, p0, p1, ch, acc, act, tag in EPI:
        oh, ow = h * up + p0 + p1 - k + 1, w * up + p0 + p1 - k + 1
        torch.manual_seed(0)
        x = torch.randn(major, h, w, device=dev)
        kk = torch.rand(k, k, device=dev)
        y0 = torch.randn(major, oh, ow, device=dev)
        ref = torch.randn(major, oh, ow, device=dev) if act else None
        gb = torch.empty(ch, device=dev)
        gbytes = 4.0 * (x.numel() + y0.numel() * (1 + acc + act)) / 1e9
        best, outs = [1e9] * len(libs), []
        for rnd in range(2):
            for i, lib in enumerate(libs):
                nws = lib.query("upfirdn2d_epilogue_workspace", major, oh, ow, ch, up) if act else 0
                ws = torch.empty(max(nws, 1), device=dev)
                y = y0.clone()
                fn = lambda: lib.call("upfirdn2d_epilogue_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), major, h, w, k, k, up, p0, p1, p0,
                                      p1, L.ptr(ref), 0.2, 2 ** 0.5, gb.data_ptr() if act else None, ch, acc, ws.data_ptr(), nws, st)
                if rnd == 0:
                    fn()
                    outs.append((y.clone(), gb.clone()))
                best[i] = min(best[i], timeit(fn))
        same = [bool(torch.equal(o[0], outs[0][0]) and (not act or torch.equal(o[1], outs[0][1]))) for o in outs]
        print("%-40s " % tag + " ".join("%9.2f%s" % (gbytes / t, " " if ok else "*") for t, ok in zip(best, same)), flush=True)
    # plain decimate x2 / zero-insert x2 (sae_upfirdn2d_f32, 4 x 4 taps): the ResBlock skip path and its gradient
    for planes, h, down, tag in [(5120, 257, 2, "decimate x2 257^2 n40"), (10240, 129, 2, "decimate x2 129^2 n40"),
                                 (12288, 129, 2, "decimate x2 129^2 Dp"), (5120, 128, 1, "zero-insert x2 128^2 n40 (plain)")]:
        up = 2 if down == 1 else 1
        p0, p1 = (2, 1) if up == 2 else (1, 1)
        oh = (h * up + p0 + p1 - 4 + down) // down
        x = torch.randn(planes, h, h, device=dev)
        kk = torch.rand(4, 4, device=dev)
        gbytes = 4.0 * (x.numel() + planes * oh * oh) / 1e9
        best, outs = [1e9] * len(libs), []
        for rnd in range(2):
            for i, lib in enumerate(libs):
                y = torch.empty(planes, oh, oh, device=dev)
                fn = lambda: lib.call("upfirdn2d_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, h, h, 1, 4, 4, up, up, down, down,
                                      p0, p1, p0, p1, st)
                if rnd == 0:
                    fn()
                    outs.append(y.clone())
                best[i] = min(best[i], timeit(fn))
        same = [bool(torch.equal(o, outs[0])) for o in outs]
        print("%-40s " % tag + " ".join("%9.2f%s" % (gbytes / t, " " if ok else "*") for t, ok in zip(best, same)), flush=True)
    for planes, h, w, ch, tag in FWD:
        oh, ow = h - 1, w - 1
        x = torch.randn(planes, h, w, device=dev)
        kk = torch.rand(4, 4, device=dev)
        nz = torch.randn(planes // ch, oh, ow, device=dev)
        nw, b = torch.full((1,), 0.3, device=dev), torch.randn(ch, device=dev)
        gbytes = 4.0 * (x.numel() + planes * oh * ow) / 1e9
        best, outs = [1e9] * len(libs), []
        for rnd in range(2):
            for i, lib in enumerate(libs):
                y = torch.empty(planes, oh, ow, device=dev)
                fn = lambda: lib.call("upfirdn2d_noise_bias_act_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, h, w, 4, 4, 1, 1,
                                      1, 1, nz.data_ptr(), nw.data_ptr(), b.data_ptr(), ch, 0.2, 2 ** 0.5, st)
                if rnd == 0:
                    fn()
                    outs.append(y.clone())
                best[i] = min(best[i], timeit(fn))
        same = [bool(torch.equal(o, outs[0])) for o in outs]
        print("%-40s " % tag + " ".join("%9.2f%s" % (gbytes / t, " " if ok else "*") for t, ok in zip(best, same)), flush=True)


if __name__ == "__main__":
    main()
