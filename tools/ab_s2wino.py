"""Same-box A/B of the stride-2 3x3 family: the direct MFMA kernels (sae_conv2d_{fwd,dgrad,wgrad}_f32 on prepared weights) against
the polyphase minimal-filtering kernels (sae_s2wino_*) on the step's shapes.  Prints ms and TFLOP/s of ALGORITHMIC (direct) FLOPs
per launch, the result's deviation from the direct kernel's, and the ratio.
   python tools/ab_s2wino.py [dgrad|fwd|wgrad ...] [--json out.json]"""
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

# n, c (large-side channels), m (small-side channels), h = w of the SMALL side, tag: the stride-2 layers of the church256 iteration
# (profiles/r6_roofline_by_shape_church256.txt) + the other presets' big ones
SHAPES = [
    (40, 128, 256, 128, "D 128->256 @257 n40"),
    (24, 128, 256, 128, "D 128->256 @257 n24"),
    (40, 256, 512, 64, "D 256->512 @129 n40"),
    (24, 256, 512, 64, "D 256->512 @129 n24"),
    (40, 512, 512, 32, "D 512->512 @65 n40"),
    (24, 512, 512, 32, "D 512->512 @65 n24"),
    (40, 512, 512, 16, "D 512->512 @33 n40"),
    (40, 512, 512, 8, "D 512->512 @17 n40"),
    (16, 256, 128, 128, "G up 128<-256 @257 n16"),
    (16, 512, 256, 64, "G up 256<-512 @129 n16"),
    (16, 512, 512, 32, "G up 512<-512 @65 n16"),
    (16, 512, 512, 16, "G up 512<-512 @33 n16"),
    (8, 256, 128, 128, "G up 128<-256 @257 n8"),
    (384, 32, 64, 64, "Dp 32->64 @129 n384"),
    (384, 64, 128, 32, "Dp 64->128 @65 n384"),
    (384, 128, 256, 16, "Dp 128->256 @33 n384"),
    (384, 256, 384, 8, "Dp 256->384 @17 n384"),
    (384, 384, 384, 4, "Dp 384->384 @9 n384"),
]


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


def main():
    argv = sys.argv[1:]
    opts = {}
    for flag in ("--json", "--lib", "--only"):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    mod = "--mod" in argv          # the modulated form: g_scale [n, m] and out_scale [n, c] on the polyphase side (timing only)
    if mod:
        argv.remove("--mod")
    ops = argv or ["dgrad"]
    out_json = opts.get("--json")
    lib = L.get()
    plib = L.SaeLibrary(os.path.abspath(opts["--lib"])) if "--lib" in opts else lib      # the polyphase side's library (a variant)
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = []
    for n, c, m, s, tag in SHAPES:
        if "--only" in opts and opts["--only"] not in tag:
            continue
        big = 2 * s + 1
        if n * max(c * big * big, m * s * s) * 4 >= (1 << 31):
            continue
        d = H.conv_desc(n, c, big, big, m, 3, 2, 0)
        flops = 2.0 * n * m * s * s * c * 9
        torch.manual_seed(1)
        x = torch.randn(n, c, big, big, device=dev)
        gy = torch.randn(n, m, s, s, device=dev)
        wt = torch.randn(m, c, 3, 3, device=dev) * 0.05
        for op in ops:
            if op == "dgrad":
                n_ws = lib.query("conv2d_workspace", C.byref(d), 1)
                ws = torch.empty(max(n_ws, 1), device=dev)
                out_a = torch.empty(n, c, big, big, device=dev)
                out_b = torch.full((n, c, big, big), float("nan"), device=dev)
                # prepared weights for the direct kernel (what the step does)
                lay, fl = C.c_int64(0), C.c_int64(0)
                lib.call("conv2d_wprep_query", C.byref(d), None, 1, C.byref(fl), C.byref(lay))
                if fl.value > 0:
                    prep = torch.empty(fl.value, device=dev)
                    lib.call("conv2d_wprep_f32", wt.data_ptr(), C.byref(d), None, 1, 1.0, prep.data_ptr(), fl.value, st)
                    d.prepped, d.prepped_floats, d.prepped_layout = prep.data_ptr(), fl.value, lay.value
                uf = torch.empty(plib.query("s2wino_weights_floats", c, m), device=dev)
                plib.call("s2wino_weights_f32", wt.data_ptr(), None, None, uf.data_ptr(), c, m, 9, c * 9, 1, 1.0, st)

                def direct():
                    lib.call("conv2d_dgrad_f32", gy.data_ptr(), wt.data_ptr(), out_a.data_ptr(), C.byref(d), 1.0, ws.data_ptr(), n_ws, st)

                gs = torch.rand(n, m, device=dev) + 0.5 if mod else None
                osc = torch.rand(n, c, device=dev) + 0.5 if mod else None

                def poly():
                    plib.call("s2wino_dgrad_f32", gy.data_ptr(), L.ptr(gs), uf.data_ptr(), L.ptr(osc), out_b.data_ptr(), n, m, c, s, s, st)
            else:
                raise SystemExit("unknown op " + op)
            iters = max(3, min(30, int(0.05 / (flops / 100e12))))
            ta = tb = 1e9
            for _ in range(2):
                ta = min(ta, timeit(direct, iters))
                tb = min(tb, timeit(poly, iters))
            err = float((out_a - out_b).abs().max() / out_a.abs().max())
            row = dict(op=op, tag=tag, n=n, c=c, m=m, small=s, direct_ms=ta, poly_ms=tb, direct_tf=flops / ta / 1e9, poly_tf=flops / tb / 1e9,
                       ratio=ta / tb, rel_dev=err)
            rows.append(row)
            print("%-6s %-26s direct %7.3f ms %6.1f TF/s   polyphase %7.3f ms %6.1f TF/s (%.3f of 157.3 executed)   x%.3f   dev %.1e" % (
                op, tag, ta, row["direct_tf"], tb, row["poly_tf"], row["poly_tf"] * 25 / 36 / 157.3, row["ratio"], err), flush=True)
            d.prepped, d.prepped_floats, d.prepped_layout = None, 0, 0
    if out_json:
        with open(out_json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
