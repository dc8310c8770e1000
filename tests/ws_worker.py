"""Subprocess body of tests/test_ws_gather.py: plain 3x3 stride-1 launches that take the 64 x 256 tile (33 ... 64 output channels, or
maps at least 128 wide) through the library with SAE_WS as the parent set it (read once per process); one sha256 per result, so the
parent can require conv_igemm_ws_kernel (producer / consumer waves, LDS-DMA staging) to be BIT-identical to conv_igemm_kernel.
    python tests/ws_worker.py emu|gpu"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import abi_harness as H  # noqa: E402
from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary  # noqa: E402

# n, c, h, w, m, pad: channel tails (rows of zeros in the last chunk), valid padding with a partial tile column, 32-wide tile rows,
# a batch tail, one chunk only, M tail; the data gradient takes the kernel when c is in 33 ... 64
EMU = [(2, 20, 16, 16, 40, 1), (1, 12, 20, 36, 64, 0), (2, 8, 8, 32, 40, 1), (1, 40, 16, 16, 36, 1), (3, 5, 16, 16, 33, 1)]
GPU = EMU + [(4, 128, 128, 128, 128, 1), (2, 64, 256, 256, 64, 1), (3, 204, 128, 128, 102, 1), (16, 128, 256, 256, 128, 1)]


def main(which):
    if which == "emu":
        from emu import build_emu
        lib, dev, cases = SaeLibrary(build_emu.build(), prefix="sae_", device_only=False), None, EMU
    else:
        from tuning import build_tuning
        lib, dev, cases = SaeLibrary(build_tuning.build()), "cuda:0", GPU
    rng = np.random.default_rng(11)
    for n, c, h, w, m, p in cases:
        d = H.conv_desc(n, c, h, w, m, 3, 1, p)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = rng.standard_normal((m, c, 3, 3)).astype(np.float32)
        gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
        b = rng.standard_normal(m).astype(np.float32)
        outs = [("fwd", H.conv(lib, 0, d, x, wt, gy.shape, alpha=0.37, device=dev)),
                ("dgrad", H.conv(lib, 1, d, gy, wt, x.shape, alpha=0.37, device=dev)),
                ("fwd+bias+lrelu", H.conv_bias_act(lib, d, x, wt, b, alpha=0.11, device=dev))]
        for tag, o in outs:
            assert not np.isnan(o).any(), (tag, n, c, h, w, m, p)
            print("%s %s %s" % ((n, c, h, w, m, p), tag, hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest()))
    print("ws-done")


if __name__ == "__main__":
    main(sys.argv[1])
