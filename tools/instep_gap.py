"""Why do the big 3x3 stride-1 convs run ~7 % slower inside the train step than in tools/ab_conv.py?  Times the same launch
(a) as the micro-benchmark does (n = 16, back to back, same buffers), (b) at the step's batch (n = 40), (c) with the fused
bias + leaky-ReLU epilogue, (d) with a 2 GiB stream between launches (cold caches / TLBs), (e) rotating through 12 distinct
input / output buffers.   python tools/instep_gap.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from swapping_autoencoder_pytorch_amd import hip_lib as L  # noqa: E402
import abi_harness as H  # noqa: E402

dev = torch.device("cuda:0")
lib = L.get()
st = lambda: torch.cuda.current_stream(dev).cuda_stream


def run(n, c, hw, m, fused=False, flush=False, rotate=1, iters=8):
    d = H.conv_desc(n, c, hw, hw, m, 3, 1, 1)
    xs = [torch.randn(n, c, hw, hw, device=dev) for _ in range(rotate)]
    ys = [torch.empty(n, m, hw, hw, device=dev) for _ in range(rotate)]
    w = torch.randn(m, c, 3, 3, device=dev)
    b = torch.randn(m, device=dev)
    nws = lib.query("conv2d_workspace", C.byref(d), 0)
    ws = torch.empty(max(nws, 1), device=dev)
    junk = torch.empty(512 << 20, device=dev) if flush else None
    flops = 2.0 * n * m * hw * hw * c * 9

    def call(i):
        x, y = xs[i % rotate], ys[i % rotate]
        if fused:
            lib.call("conv2d_fwd_bias_act_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), 1.0, 0.2,
                     2 ** 0.5, ws.data_ptr(), nws, st())
        else:
            lib.call("conv2d_fwd_f32", x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 1.0, ws.data_ptr(), nws, st())
    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    total = 0.0
    for i in range(iters):
        if flush:
            junk.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(i); e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
    return flops / (total / iters) / 1e9


for (c, hw, m) in [(128, 256, 128), (256, 128, 256), (512, 64, 512)]:
    print("3x3 s1 %d->%d @%d:  n16 %.1f | n40 %.1f | n40 fused bias+lrelu %.1f | n40 after a 2 GiB stream %.1f | n40 rotating 6 buffers %.1f   TFLOP/s"
          % (c, m, hw, run(16, c, hw, m), run(40, c, hw, m), run(40, c, hw, m, fused=True), run(40, c, hw, m, flush=True),
             run(40, c, hw, m, rotate=6)), flush=True)
