"""Fixed cost per workgroup of wino_fused_kernel: the one-kernel Winograd conv at 128 output channels on 256 x 256 maps with 8 .. 256
input channels (1 .. 32 passes per workgroup): ms per launch and us per round of workgroups.   python tools/wf_fixed_cost.py [variant]"""
import os, sys, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi_harness as H
from swapping_autoencoder_pytorch_amd import hip_lib
lib = hip_lib.SaeLibrary(os.path.join(ROOT, "tools", "variants", sys.argv[1] + ".so")) if len(sys.argv) > 1 else hip_lib.get()
print("library:", sys.argv[1] if len(sys.argv) > 1 else "product")
dev = "cuda:0"
st = torch.cuda.current_stream(dev).cuda_stream
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n, m, side = 16, 128, 256
for c in (8, 16, 32, 64, 128, 256):
    x = torch.randn(n, c, side, side, device=dev)
    w = torch.randn(m, c, 3, 3, device=dev)
    y = torch.empty(n, m, side, side, device=dev)
    u = torch.empty(lib.query("wino_fused_weights_floats", m, c), device=dev)
    lib.call("wino_fused_weights_f32", w.data_ptr(), None, None, u.data_ptr(), m, c, c * 9, 9, 0, 1.0, st)
    f = lambda: lib.call("wino_fused_conv_f32", x.data_ptr(), None, u.data_ptr(), None, None, None, None, y.data_ptr(), n, c, m, side, side, 1, 0, 0.0, 1.0, st)
    t = timed(f)
    wgs = n * (side // 2) ** 2 // 64 * (m // 64)
    print("C=%4d chunks=%3d  %.3f ms  per workgroup round (%d WGs / 256 CUs = %.1f rounds): %.2f us" % (c, c // 8, t, wgs, wgs / 256, t * 1e3 / (wgs / 256)))
