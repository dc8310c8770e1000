"""Is the one-kernel Winograd forward's bimodal launch time (1.19 / 1.35 ms at 512 -> 512 @64x64, B = 16, between allocations) a function of
WHERE x, the prepared weights and y lie?  One arena, x fixed, y (and then x) moved by a sweep of byte offsets; HIP events around 10 launches.
    python tools/wf_offsets.py > gpurun_out/wf_offsets.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402


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
    lib = hip_lib.get()
    dev = torch.device("cuda:0")
    for n, c, side in ((16, 512, 64), (16, 256, 128), (16, 128, 256)):
        numel = n * c * side * side
        slack = (64 << 20) // 4
        arena = torch.empty(2 * numel + 3 * slack, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev)
        u = torch.empty(lib.query("wino_fused_weights_floats", c, c), dtype=torch.float32, device=dev)
        lib.call("wino_fused_weights_f32", w.data_ptr(), None, None, u.data_ptr(), c, c, c * 9, 9, 0, 1.0 / (c * 9) ** 0.5, lib.stream(w))
        stream = lib.stream(w)
        base = arena.data_ptr()
        base += (-base) % (2 << 20)                      # 2 MiB aligned start
        ex = 32.0 * n * c * c * (side // 2) ** 2
        print("# %d -> %d @%d, B = %d: x at a 2 MiB boundary, y at x + bytes(x) + offset; ms and fraction of the fp32 MFMA peak (executed FLOPs)" % (c, c, side, n))
        xs = torch.randn(numel, device=dev)
        for xoff in (0, 4096, 1 << 20):
            xp = base + xoff
            xv = torch.empty(0)
            # fill x through a view of the arena
            start = (xp - arena.data_ptr()) // 4
            arena[start:start + numel].copy_(xs)
            row = []
            for yoff in (0, 256, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 65536, 3 << 20, 5 << 20, 16 << 20, (16 << 20) + 4096 * 3):
                yp = xp + numel * 4 + yoff
                assert yp + numel * 4 <= arena.data_ptr() + arena.numel() * 4
                t = timed(lambda: lib.call("wino_fused_conv_f32", xp, None, u.data_ptr(), None, None, None, None, yp, n, c, c, side, side, 1, 0,
                                           0.2, 1.0, stream))
                row.append("%s: %.3f (%.3f)" % (yoff, t, ex / t / 1e9 / 157.3))
            print("x + %-8d | %s" % (xoff, " | ".join(row)), flush=True)
        del arena


if __name__ == "__main__":
    main()
