"""Launch a fixed set of hot kernels a few times each (for rocprofv3 --pmc passes):
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -d out -- python tools/pmc_kernels.py"""
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


def conv(n, c, h, w, m, k, s, p, ops=(0, 1, 2), reps=3):
    d = H.conv_desc(n, c, h, w, m, k, s, p)
    x = torch.randn(n, c, h, w, device=dev)
    wt = torch.randn(m, c, k, k, device=dev)
    y = torch.empty(n, m, d.oh, d.ow, device=dev)
    gy = torch.randn_like(y)
    gx = torch.empty_like(x)
    gw = torch.empty_like(wt)
    for op, (a, b, o) in [(0, (x, wt, y)), (1, (gy, wt, gx)), (2, (x, gy, gw))]:
        if op not in ops:
            continue
        nws = lib.query("conv2d_workspace", C.byref(d), op)
        ws = torch.empty(max(nws, 1), device=dev)
        for _ in range(reps):
            lib.call(H.OPS[op], a.data_ptr(), b.data_ptr(), o.data_ptr(), C.byref(d), 1.0, ws.data_ptr(), nws, st())
    torch.cuda.synchronize()


def blur(planes, h, w, k, pad, reps=3):
    x = torch.randn(planes, h, w, device=dev)
    kk = torch.ones(k, k, device=dev) / (k * k)
    y = torch.empty(planes, h + 2 * pad - k + 1, w + 2 * pad - k + 1, device=dev)
    for _ in range(reps):
        lib.call("upfirdn2d_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, h, w, 1, k, k, 1, 1, 1, 1, pad, pad,
                 pad, pad, st())
    torch.cuda.synchronize()


def calibrate():
    """Known byte counts for the FETCH_SIZE / WRITE_SIZE calibration the guide asks for (MI355X_MICROARCH.md, HBM):
    a 1 GiB ATen copy (16 B per lane: FETCH_SIZE is expected to report half) and a 1 GiB copy through the blur kernel
    with a single unit tap (4 B per lane, the access width of the conv kernels' patch staging)."""
    a = torch.randn(256 * 1024 * 1024, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    planes, h, w = 1024, 512, 512
    x = a.view(planes, h, w)
    y = b.view(planes, h, w)
    one = torch.ones(1, 1, device=dev)
    for _ in range(3):
        lib.call("upfirdn2d_f32", x.data_ptr(), one.data_ptr(), y.data_ptr(), planes, h, w, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, st())
    torch.cuda.synchronize()


if __name__ == "__main__":
    if "--tr" in sys.argv:      # the transposed gather alone (stride-2 data gradient), plain and at the generator's batch
        calibrate()
        conv(16, 128, 257, 257, 256, 3, 2, 0, ops=(1,))
        conv(16, 256, 129, 129, 512, 3, 2, 0, ops=(1,))
        sys.exit(0)
    if "--s2" in sys.argv:      # the stride-2 family at the step's three wide shapes: gather / transposed gather / weight gradient
        conv(16, 128, 257, 257, 256, 3, 2, 0)
        conv(16, 256, 129, 129, 512, 3, 2, 0)
        conv(16, 512, 65, 65, 512, 3, 2, 0)
        sys.exit(0)
    calibrate()
    conv(16, 128, 256, 256, 128, 3, 1, 1)          # igemm s1 / dgrad s1 / wgrad s1
    conv(16, 128, 257, 257, 256, 3, 2, 0)          # igemm s2 / tr / wgrad s2
    conv(16, 512, 16, 16, 512, 3, 1, 1, ops=(0,))  # split-K tail
    conv(16, 128, 128, 128, 256, 1, 1, 0)          # 1x1 family
    blur(2048, 256, 256, 4, 2)
    blur(2048, 257, 257, 4, 1)
