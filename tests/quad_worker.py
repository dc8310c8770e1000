"""Subprocess body of tests/test_quad_paths.py: runs a fixed set of conv launches through the library with whatever staging
knobs the parent set (SAE_IGEMM_QUAD / SAE_WGRAD_QUAD / SAE_IGEMM_VEC_STORE / SAE_CONV_THIN are read once per process) and
prints one sha256 per result, so that the parent can require the quad-staged kernels to be BIT-identical to the dword ones.
    python tests/quad_worker.py emu|gpu"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import abi_harness as H  # noqa: E402
from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary  # noqa: E402

# n, c, h, w, m, k, stride, pad: every quad path -- 3x3 s1 on the 128- and 256-pixel tiles (16- and 32-wide rows, channel
# tails, valid padding), 1x1 s1, 3x3 s2 with 2^k + 1 wide rows (shifted last quad, partial tile), thin 1x1 layers excluded
# (different arithmetic order by design, checked against the oracle in the kernel tests)
EMU = [(2, 20, 16, 16, 130, 3, 1, 1), (1, 12, 20, 36, 70, 3, 1, 0), (3, 20, 12, 12, 24, 3, 1, 1), (1, 70, 16, 16, 70, 1, 1, 0),
       (2, 12, 21, 41, 70, 3, 2, 0), (1, 6, 33, 33, 130, 3, 2, 0),
       # stride-2 data gradient through conv_igemm_tr2_kernel (input rows a multiple of 4 wide): M / channel tails, strips,
       # pad 1, 32-row tile, several images per tile
       (2, 40, 17, 33, 20, 3, 2, 0), (1, 36, 16, 24, 12, 3, 2, 1), (3, 20, 17, 17, 40, 3, 2, 0), (1, 70, 9, 65, 9, 3, 2, 0),
       # stride 2 with INTERIOR tiles and full channel blocks (137 wide: two of three tile columns touch no border; 128 x 32
       # channels): the select-free staging of the gather and of the weight gradient, next to the masked one in the same launch
       (1, 32, 9, 137, 128, 3, 2, 0), (2, 40, 13, 137, 136, 3, 2, 0)]
GPU = EMU + [(4, 128, 64, 64, 128, 3, 1, 1), (2, 512, 32, 32, 512, 3, 1, 1), (8, 32, 64, 64, 32, 3, 1, 1), (4, 64, 65, 65, 128, 3, 2, 0),
             (4, 128, 64, 64, 256, 1, 1, 0), (2, 256, 129, 129, 512, 3, 2, 0)]


def main(which):
    if which == "emu":
        from emu import build_emu
        lib, dev, cases = SaeLibrary(build_emu.build(), prefix="sae_", device_only=False), None, EMU
    else:
        # the tuning build of the same kernel sources (tests/tuning): the product library has no dispatch knobs
        from tuning import build_tuning
        lib, dev, cases = SaeLibrary(build_tuning.build()), "cuda:0", GPU
    rng = np.random.default_rng(5)
    for n, c, h, w, m, k, s, p in cases:
        d = H.conv_desc(n, c, h, w, m, k, s, p)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = rng.standard_normal((m, c, k, k)).astype(np.float32)
        gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
        b = rng.standard_normal(m).astype(np.float32)
        outs = [("fwd", H.conv(lib, 0, d, x, wt, gy.shape, alpha=0.37, device=dev)),
                ("dgrad", H.conv(lib, 1, d, gy, wt, x.shape, alpha=0.37, device=dev)),
                ("wgrad", H.conv(lib, 2, d, x, gy, wt.shape, alpha=0.37, device=dev)),
                ("fwd+bias+lrelu", H.conv_bias_act(lib, d, x, wt, b, alpha=0.11, device=dev))]
        for tag, o in outs:
            assert not np.isnan(o).any(), (tag, n, c, h, w, m, k, s, p)
            print("%s %s %s" % ((n, c, h, w, m, k, s, p), tag, hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest()))
    print("quad-done")


if __name__ == "__main__":
    main(sys.argv[1])
