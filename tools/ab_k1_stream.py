"""Same-box A/B of the streaming blur (csrc/upfirdn2d.hip: blur_stream_kernel) against the LDS-strip kernels it replaces, on the
planes of the church256 / ffhq presets: plain blur (sae_upfirdn2d_f32) and the blur + noise + bias + leaky-ReLU forward
(sae_upfirdn2d_noise_bias_act_f32).  Uses the tuning build of the kernel sources (tests/tuning), whose dispatch knob
SAE_K1_STREAM selects the kernel per call; prints TB/s of algorithmic bytes (4 x (numel_in + numel_out)).
    python tools/ab_k1_stream.py > gpurun_out/ab_k1_stream.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary  # noqa: E402
from tuning import build_tuning  # noqa: E402

lib = SaeLibrary(build_tuning.build())
dev = torch.device("cuda:0")


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


def run(planes, h, w, k, pad, act):
    x = torch.randn(planes, h, w, device=dev)
    kk = torch.ones(k, k, device=dev) / (k * k)
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    y = torch.empty(planes, oh, ow, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if act:
        ch = 128
        nz = torch.randn(planes // ch, oh, ow, device=dev)
        nw, b = torch.full((1,), 0.3, device=dev), torch.randn(ch, device=dev)
        fn = lambda: lib.call("upfirdn2d_noise_bias_act_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, h, w, k, k, pad, pad,
                              pad, pad, nz.data_ptr(), nw.data_ptr(), b.data_ptr(), ch, 0.2, 2 ** 0.5, st)
    else:
        fn = lambda: lib.call("upfirdn2d_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, h, w, 1, k, k, 1, 1, 1, 1, pad, pad,
                              pad, pad, st)
    gb = 4.0 * (x.numel() + y.numel()) / 1e9
    out = []
    for knob in ("0", "2"):
        os.environ["SAE_K1_STREAM"] = knob
        ms = timeit(fn)
        out.append((ms, gb / ms))
    print("%-34s planes %6d %4dx%-4d k%d pad%d   strip %.3f ms %5.2f TB/s   stream %.3f ms %5.2f TB/s   x%.2f"
          % ("blur+noise+bias+lrelu fwd" if act else "blur", planes, h, w, k, pad, out[0][0], out[0][1], out[1][0], out[1][1],
             out[0][0] / out[1][0]), flush=True)


if __name__ == "__main__":
    # the blur in front of every stride-2 conv of D / E / Dpatch (pad 2: 2^k -> 2^k + 1), B = 16 + 8 (D), 128 x 3 (Dpatch)
    for planes, hw in ((5120, 256), (3072, 256), (2048, 256), (12288, 128), (10240, 128), (4096, 128), (24576, 64), (8192, 64),
                       (49152, 32), (16384, 32)):
        run(planes, hw, hw, 4, 2, False)
    # the blur that ends the generator's upsampling convs (pad 1: 2^k + 1 -> 2^k) + noise + bias + activation
    for planes, hw in ((2048, 257), (1024, 257), (4096, 129), (2048, 129), (4096, 65), (8192, 33)):
        run(planes, hw, hw, 4, 1, True)
    # E: [1, 2, 1] blur after reflection padding; ffhq512 / ffhq1024 planes
    run(512, 259, 259, 3, 0, False)
    run(1024, 131, 131, 3, 0, False)
    run(8 * 64, 512, 512, 4, 2, False)
    run(4 * 32, 1024, 1024, 4, 2, False)
    run(8 * 64, 513, 513, 4, 1, True)
