"""Per-kernel micro-benchmark at the church-preset shapes: reports GB/s of
the HBM-bound ops against their algorithmic bytes and TFLOP/s of the MFMA convs.
Run on the GPU box:  python tools/kernel_bench.py [--quick] > gpurun_out/kernel_bench.jsonl"""
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
if os.environ.get("KB_LIBRARY"):        # A/B against a variant build (tools/variants/*.so); default: the product library
    L._LIB = L.SaeLibrary(os.path.abspath(os.environ["KB_LIBRARY"]))
lib = L.get()
quick = "--quick" in sys.argv


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def stream():
    return torch.cuda.current_stream(dev).cuda_stream


def bench_blur(planes, h, w, k, pad):
    x = torch.randn(planes, h, w, device=dev)
    kk = torch.ones(k, k, device=dev) / (k * k)
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    y = torch.empty(planes, oh, ow, device=dev)
    fn = lambda: lib.call("upfirdn2d_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, h, w, 1, k, k, 1, 1, 1, 1,
                          pad, pad, pad, pad, stream())
    ms = timeit(fn)
    gb = 4.0 * (x.numel() + y.numel()) / 1e9
    print(json.dumps({"op": "upfirdn2d", "shape": [planes, h, w], "k": k, "pad": pad, "ms": ms, "GBps": gb / ms * 1e3}), flush=True)


def bench_bias_act(shape):
    x = torch.randn(*shape, device=dev)
    b = torch.randn(shape[1], device=dev)
    y = torch.empty_like(x)
    step = x[0, 0].numel() if x.dim() > 2 else 1
    fn = lambda: lib.call("bias_act_f32", x.data_ptr(), b.data_ptr(), None, y.data_ptr(), x.numel(), step, shape[1], 3, 0,
                          0.2, 2 ** 0.5, stream())
    ms = timeit(fn)
    print(json.dumps({"op": "bias_act_fwd", "shape": list(shape), "ms": ms, "GBps": 8.0 * x.numel() / 1e6 / ms}), flush=True)
    gx = torch.empty_like(x)
    gb = torch.empty(shape[1], device=dev)
    n = lib.query("bias_act_bwd_workspace", x.numel(), step, shape[1])
    ws = torch.empty(max(n, 1), device=dev)
    fn = lambda: lib.call("bias_act_bwd_f32", x.data_ptr(), y.data_ptr(), gx.data_ptr(), gb.data_ptr(), ws.data_ptr(), n,
                          x.numel(), step, shape[1], 0.2, 2 ** 0.5, stream())
    ms = timeit(fn)
    print(json.dumps({"op": "bias_act_bwd", "shape": list(shape), "ms": ms, "GBps": 12.0 * x.numel() / 1e6 / ms}), flush=True)


def bench_conv_fused(n, c, h, w, m, k, s, p, tag):
    """forward conv with the fused bias + leaky-ReLU epilogue (sae_conv2d_fwd_bias_act_f32)"""
    d = H.conv_desc(n, c, h, w, m, k, s, p)
    x = torch.randn(n, c, h, w, device=dev)
    wt = torch.randn(m, c, k, k, device=dev)
    b = torch.randn(m, device=dev)
    y = torch.empty(n, m, d.oh, d.ow, device=dev)
    nws = lib.query("conv2d_workspace", C.byref(d), 0)
    ws = torch.empty(max(nws, 1), device=dev)
    fn = lambda: lib.call("conv2d_fwd_bias_act_f32", x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d),
                          1.0, 0.2, 2 ** 0.5, ws.data_ptr(), nws, stream())
    ms = timeit(fn)
    fl = 2.0 * n * m * d.oh * d.ow * c * k * k
    print(json.dumps({"op": "conv_fwd_bias_act", "tag": tag, "geom": [n, c, h, w, m, k, s, p], "ms": ms,
                      "TFLOPs": fl / ms / 1e9}), flush=True)


def bench_conv(n, c, h, w, m, k, s, p, tag):
    d = H.conv_desc(n, c, h, w, m, k, s, p)
    x = torch.randn(n, c, h, w, device=dev)
    wt = torch.randn(m, c, k, k, device=dev)
    y = torch.empty(n, m, d.oh, d.ow, device=dev)
    gy = torch.randn_like(y)
    gx = torch.empty_like(x)
    gw = torch.empty_like(wt)
    flops = 2.0 * n * m * d.oh * d.ow * c * k * k
    for op, name, (a, b, o) in [(0, "fwd", (x, wt, y)), (1, "dgrad", (gy, wt, gx)), (2, "wgrad", (x, gy, gw))]:
        nws = lib.query("conv2d_workspace", C.byref(d), op)
        ws = torch.empty(max(nws, 1), device=dev)
        fn = lambda: lib.call(H.OPS[op], a.data_ptr(), b.data_ptr(), o.data_ptr(), C.byref(d), 1.0, ws.data_ptr(), nws,
                              stream())
        ms = timeit(fn, iters=5 if flops > 5e10 else 10)
        print(json.dumps({"op": "conv_" + name, "tag": tag, "geom": [n, c, h, w, m, k, s, p], "ms": ms,
                          "TFLOPs": flops / ms / 1e9, "frac_mfma_peak": flops / ms / 1e9 / 157.3,
                          "ws_MB": nws * 4 / 1e6}), flush=True)


if __name__ == "__main__":
    print(json.dumps({"device": torch.cuda.get_device_name(0)}), flush=True)
    B = 4 if quick else 16
    P = 32 if quick else 128
    bench_blur(B * 128, 256, 256, 4, 2)
    bench_blur(B * 128, 256, 256, 4, 1)
    bench_blur(B * 128, 257, 257, 4, 1)
    bench_blur(P * 32, 128, 128, 4, 2)
    bench_blur(B * 32, 259, 259, 3, 0)
    bench_blur(B * 512, 32, 32, 4, 2)
    bench_blur(P * 384, 8, 8, 4, 2)
    bench_bias_act((B, 128, 256, 256))
    bench_bias_act((P, 32, 128, 128))
    bench_bias_act((B, 512, 32, 32))
    bench_bias_act((P, 384, 8, 8))
    bench_bias_act((P, 2048))
    bench_conv(B, 128, 256, 256, 128, 3, 1, 1, "D 3x3 s1 128@256")
    bench_conv(B, 128, 257, 257, 256, 3, 2, 0, "D 3x3 s2 128->256")
    bench_conv(B, 128, 255, 255, 256, 1, 2, 0, "D 1x1 s2 skip")
    bench_conv(B, 512, 64, 64, 512, 3, 1, 1, "D 3x3 s1 512@64")
    bench_conv(B, 512, 16, 16, 512, 3, 1, 1, "D 3x3 s1 512@16")
    bench_conv(B, 512, 4, 4, 512, 3, 1, 1, "D 3x3 s1 512@4")
    bench_conv(P, 32, 128, 128, 32, 3, 1, 1, "Dp 3x3 s1 32@128")
    bench_conv(P, 64, 64, 64, 64, 3, 1, 1, "Dp 3x3 s1 64@64")
    bench_conv(P, 256, 16, 16, 256, 3, 1, 1, "Dp 3x3 s1 256@16")
    bench_conv(P, 384, 4, 4, 384, 3, 1, 1, "Dp 3x3 s1 384@4")
    bench_conv(P, 3, 128, 128, 32, 3, 1, 1, "Dp stem 3->32")
    bench_conv(B, 512, 64, 64, 256, 1, 1, 0, "G 1x1 512->256@64")
    bench_conv(B, 128, 257, 257, 256, 3, 2, 0, "G convT 256->128 (as dgrad of 128->256 s2)")
