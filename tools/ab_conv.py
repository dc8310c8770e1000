"""A/B timing of library variants (tools/build_variant.sh) on the conv shapes that carry the step:
   python tools/ab_conv.py base ig1 ig2 ...        (names under tools/variants/, or "product" for csrc/libsae_hip.so)
Every shape is timed on every variant in turn, twice round-robin, best of the two; TFLOP/s per variant side by side.
Also checks that every variant's output is bit-identical to the first one's."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from swapping_autoencoder_pytorch_amd import hip_lib as L  # noqa: E402
import abi_harness as H  # noqa: E402

dev = torch.device("cuda:0")

SHAPES = [  # n, c, h, w, m, k, s, p, tag
    (16, 128, 256, 256, 128, 3, 1, 1, "s1 128@256"),
    (16, 256, 128, 128, 256, 3, 1, 1, "s1 256@128"),
    (16, 512, 64, 64, 512, 3, 1, 1, "s1 512@64"),
    (16, 512, 32, 32, 512, 3, 1, 1, "s1 512@32"),
    (16, 512, 16, 16, 512, 3, 1, 1, "s1 512@16"),
    (128, 32, 128, 128, 32, 3, 1, 1, "Dp s1 32@128"),
    (128, 64, 64, 64, 64, 3, 1, 1, "Dp s1 64@64"),
    (128, 128, 32, 32, 128, 3, 1, 1, "Dp s1 128@32"),
    (16, 128, 257, 257, 256, 3, 2, 0, "s2 128->256@257"),
    (16, 256, 129, 129, 512, 3, 2, 0, "s2 256->512@129"),
    (16, 512, 65, 65, 512, 3, 2, 0, "s2 512->512@65"),
    (128, 32, 129, 129, 64, 3, 2, 0, "Dp s2 32->64@129"),
    (16, 128, 256, 256, 256, 1, 1, 0, "1x1 128->256@256"),
    (16, 512, 64, 64, 256, 1, 1, 0, "1x1 512->256@64"),
    # mid-size launches: a few rounds of workgroups (the image discriminator's batches of 24 / 40)
    (24, 512, 32, 32, 512, 3, 1, 1, "s1 512@32 n24"),
    (40, 512, 32, 32, 512, 3, 1, 1, "s1 512@32 n40"),
    (24, 512, 64, 64, 512, 3, 1, 1, "s1 512@64 n24"),
    (24, 512, 65, 65, 512, 3, 2, 0, "s2 512->512@65 n24"),
    # the small 2^k + 1 grids of the stride-2 data gradients (q grids 33, 17 wide)
    (16, 512, 33, 33, 512, 3, 2, 0, "s2 512->512@33"),
    (40, 512, 65, 65, 512, 3, 2, 0, "s2 512->512@65 n40"),
    (128, 64, 65, 65, 128, 3, 2, 0, "Dp s2 64->128@65"),
    (128, 128, 33, 33, 256, 3, 2, 0, "Dp s2 128->256@33"),
    # Dpatch's 384- / 768-channel tail with the discriminator step's 384 crops: tile counts that do not divide the CUs
    (384, 256, 17, 17, 384, 3, 2, 0, "Dp3 s2 256->384@17"),
    (384, 384, 8, 8, 384, 3, 1, 1, "Dp3 s1 384@8"),
    (384, 384, 9, 9, 384, 3, 2, 0, "Dp3 s2 384@9"),
    (384, 384, 4, 4, 384, 3, 1, 1, "Dp3 s1 384@4"),
    (384, 384, 4, 4, 768, 3, 1, 1, "Dp3 s1 384->768@4"),
    (384, 768, 4, 4, 384, 3, 1, 0, "Dp3 s1 768->384@4"),
    (40, 128, 128, 128, 256, 1, 1, 0, "1x1 128->256@128 n40"),
    (40, 256, 64, 64, 512, 1, 1, 0, "1x1 256->512@64 n40"),
    (40, 512, 32, 32, 512, 1, 1, 0, "1x1 512->512@32 n40"),
    (24, 512, 16, 16, 512, 3, 1, 1, "s1 512@16 n24"),
    (24, 512, 8, 8, 512, 3, 1, 1, "s1 512@8 n24"),
    (40, 512, 4, 4, 512, 3, 1, 1, "s1 512@4 n40"),
]


def libpath(name):
    if name == "product":
        return L.DEFAULT_LIBRARY
    if name == "tuning":       # the -DSAE_TUNING build of the current sources (environment knobs select the kernels)
        return os.path.join(ROOT, "tests", "tuning", "libsae_hip_tuning.so")
    if os.sep in name:
        return os.path.abspath(name)
    return os.path.join(ROOT, "tools", "variants", name + ".so")


def timeit(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def clock_mode():
    """python tools/ab_conv.py --clock: the shader clock the fp32 gather kernels actually run at (variant `clk`, built with
    -DSAE_CLOCK_PROBE): per op, TFLOP/s from events, MHz from the kernels' own s_memtime / s_memrealtime, and the rate
    re-expressed as a fraction of 1024 SIMDs x 64 FLOP/cycle at THAT clock."""
    lib = L.SaeLibrary(libpath("clk"))
    raw = C.CDLL(libpath("clk"))
    raw.sae_debug_clock_probe.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 10)()
    stream = lambda: torch.cuda.current_stream(dev).cuda_stream
    print("%-20s %-6s %9s %9s %9s %9s   %s" % ("shape", "op", "TFLOP/s", "of 157.3", "MHz", "of peak@MHz",
          "wave-0 time of conv_igemm_kernel / conv_wgrad_kernel: prologue | MFMA | barrier | LDS stores | barrier | load issue | epilogue (%)"))
    only = [a[2:] for a in sys.argv[1:] if a.startswith("--") and a != "--clock"]
    for (n, c, h, w, m, k, s, p, tag) in SHAPES:
        if only and not any(o in tag for o in only):
            continue
        d = H.conv_desc(n, c, h, w, m, k, s, p)
        x = torch.randn(n, c, h, w, device=dev)
        wt = torch.randn(m, c, k, k, device=dev)
        gy = torch.randn(n, m, d.oh, d.ow, device=dev)
        flops = 2.0 * n * m * d.oh * d.ow * c * k * k
        for op, oname in [(0, "fwd"), (1, "dgrad"), (2, "wgrad")]:
            a, b = [(x, wt), (gy, wt), (x, gy)][op]
            o = torch.empty([(n, m, d.oh, d.ow), (n, c, h, w), (m, c, k, k)][op], device=dev)
            nws = lib.query("conv2d_workspace", C.byref(d), op)
            ws = torch.empty(max(nws, 1), device=dev)
            fn = lambda: lib.call(H.OPS[op], a.data_ptr(), b.data_ptr(), o.data_ptr(), C.byref(d), 1.0, ws.data_ptr(), nws,
                                  stream())
            for _ in range(8):
                fn()                                  # ~20 ms of the same kernel: let the clock settle
            torch.cuda.synchronize()
            raw.sae_debug_clock_probe(buf, 1)
            ms = timeit(fn, 10)
            raw.sae_debug_clock_probe(buf, 1)
            mhz = buf[0] / (buf[1] / 100e6) / 1e6 if buf[1] else 0.0
            tf = flops / ms / 1e9
            ph = [buf[3 + i] for i in range(7)]
            tot = float(sum(ph)) or 1.0
            print("%-20s %-6s %9.1f %9.3f %9.0f %9.3f   %s   (%.0f kcycles per workgroup)" % (
                tag, oname, tf, tf / 157.3, mhz, tf / (1024 * 64 * mhz * 1e-6) if mhz else 0.0,
                " | ".join("%4.1f" % (100.0 * v / tot) for v in ph), buf[0] / max(buf[2], 1) / 1e3), flush=True)


def main():
    if "--clock" in sys.argv:
        return clock_mode()
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["product"]
    only = [a[2:] for a in sys.argv[1:] if a.startswith("--") and not a.startswith("--op=")]
    want_ops = [a[5:] for a in sys.argv[1:] if a.startswith("--op=")]       # --op=wgrad: only that operation
    libs = [L.SaeLibrary(libpath(n)) for n in names]
    stream = lambda: torch.cuda.current_stream(dev).cuda_stream
    print("%-20s %-6s " % ("shape", "op") + " ".join("%9s" % n for n in names) + "   (TFLOP/s; * = differs from the first)")
    totals = [0.0] * len(names)
    for (n, c, h, w, m, k, s, p, tag) in SHAPES:
        if only and not any(o in tag for o in only):
            continue
        d = H.conv_desc(n, c, h, w, m, k, s, p)
        torch.manual_seed(1)
        x = torch.randn(n, c, h, w, device=dev)
        wt = torch.randn(m, c, k, k, device=dev)
        gy = torch.randn(n, m, d.oh, d.ow, device=dev)
        flops = 2.0 * n * m * d.oh * d.ow * c * k * k
        bias = torch.randn(m, device=dev)
        res = torch.randn(n, m, d.oh, d.ow, device=dev)
        # fwdact / fwdres (only with --op=): the forward with its fused epilogues (bias + leaky ReLU; residual merge, 1x1 only)
        for op, oname in [(0, "fwd"), (1, "dgrad"), (2, "wgrad"), (0, "fwdact"), (0, "fwdres")]:
            if (want_ops and oname not in want_ops) or (not want_ops and oname in ("fwdact", "fwdres")):
                continue
            if oname == "fwdres" and k != 1:
                continue
            a, b = [(x, wt), (gy, wt), (x, gy)][op]
            shape = [(n, m, d.oh, d.ow), (n, c, h, w), (m, c, k, k)][op]
            best = [1e9] * len(libs)
            outs = []
            for rnd in range(2):
                for i, lib in enumerate(libs):
                    o = torch.empty(shape, device=dev)
                    nws = lib.query("conv2d_workspace", C.byref(d), op)
                    ws = torch.empty(max(nws, 1), device=dev)
                    if oname == "fwdact":
                        fn = lambda: lib.call("conv2d_fwd_bias_act_f32", a.data_ptr(), b.data_ptr(), bias.data_ptr(), o.data_ptr(),
                                              C.byref(d), 1.0, 0.2, 2 ** 0.5, ws.data_ptr(), nws, stream())
                    elif oname == "fwdres":
                        fn = lambda: lib.call("conv2d_fwd_residual_f32", a.data_ptr(), b.data_ptr(), res.data_ptr(), o.data_ptr(),
                                              C.byref(d), 1.0, 2 ** -0.5, ws.data_ptr(), nws, stream())
                    else:
                        fn = lambda: lib.call(H.OPS[op], a.data_ptr(), b.data_ptr(), o.data_ptr(), C.byref(d), 1.0,
                                              ws.data_ptr(), nws, stream())
                    best[i] = min(best[i], timeit(fn, 5 if flops > 5e10 else 10))
                    if rnd == 0:
                        outs.append(o)
            same = [bool(torch.equal(o, outs[0])) for o in outs]
            for i in range(len(libs)):
                totals[i] += best[i]
            print("%-20s %-6s " % (tag, oname) + " ".join("%8.1f%s" % (flops / t / 1e9, " " if ok else "*")
                                                          for t, ok in zip(best, same)), flush=True)
    print("%-27s " % "sum of best ms" + " ".join("%9.3f" % t for t in totals))
    print(json.dumps({"variants": names, "sum_ms": totals}))


if __name__ == "__main__":
    main()
