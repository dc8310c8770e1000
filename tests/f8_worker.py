"""Subprocess body of tests/test_f8_gather.py: with SAE_F8_MIN_TILES=1 (set by the parent before this process loads the
library) every 3x3 stride-1 launch with more than 64 output channels takes conv_igemm_f8_kernel, so small shapes
exercise its tails, several-images-per-tile patches, split-K slabs, the fused epilogue and the modulated variant.
    python tests/f8_worker.py emu|gpu"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import abi_harness as H  # noqa: E402
from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary  # noqa: E402

CASES = [(2, 8, 8, 8, 70, 1, False), (3, 10, 4, 4, 70, 1, False), (1, 9, 36, 33, 130, 0, False), (2, 64, 8, 8, 100, 1, True),
         (1, 5, 16, 40, 256, 1, False), (2, 72, 6, 6, 100, 1, True), (1, 17, 40, 40, 128, 1, False)]


def main(which):
    assert os.environ.get("SAE_F8_MIN_TILES") == "1"
    oracle = SaeLibrary(os.path.join(ROOT, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
    if which == "emu":
        from emu import build_emu
        lib, dev = SaeLibrary(build_emu.build(), prefix="sae_", device_only=False), None
    else:
        # the tuning build of the same kernel sources (tests/tuning): the product library does not contain this kernel
        from tuning import build_tuning
        lib, dev = SaeLibrary(build_tuning.build()), "cuda:0"
    rng = np.random.default_rng(23)
    for n, c, h, w, m, p, cm in CASES:
        d = H.conv_desc(n, c, h, w, m, 3, 1, p, cm)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = rng.standard_normal((c, m, 3, 3) if cm else (m, c, 3, 3)).astype(np.float32)
        gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
        b = rng.standard_normal(m).astype(np.float32)
        xs = (1 + 0.5 * rng.standard_normal((n, c))).astype(np.float32)
        ys = (1 + 0.5 * rng.standard_normal((n, m))).astype(np.float32)
        wm = rng.uniform(0.5, 2, m).astype(np.float32)
        checks = [("fwd", H.conv(lib, 0, d, x, wt, gy.shape, alpha=0.37, device=dev), H.conv(oracle, 0, d, x, wt, gy.shape, alpha=0.37))]
        if c > 64:       # the data gradient produces c channels: f8 only when they exceed 64
            checks.append(("dgrad", H.conv(lib, 1, d, gy, wt, x.shape, alpha=0.37, device=dev), H.conv(oracle, 1, d, gy, wt, x.shape, alpha=0.37)))
            checks.append(("dgrad mod", H.modconv(lib, 1, d, gy, wt, x.shape, y_scale=ys, alpha=0.3, device=dev),
                           H.modconv(oracle, 1, d, gy, wt, x.shape, y_scale=ys, alpha=0.3)))
        if not cm:
            checks.append(("fwd+bias+lrelu", H.conv_bias_act(lib, d, x, wt, b, alpha=0.11, device=dev), H.conv_bias_act(oracle, d, x, wt, b, alpha=0.11)))
        checks.append(("fwd mod", H.modconv(lib, 0, d, x, wt, gy.shape, x_scale=xs, wm_scale=wm, alpha=0.3, device=dev),
                       H.modconv(oracle, 0, d, x, wt, gy.shape, x_scale=xs, wm_scale=wm, alpha=0.3)))
        for tag, e, o in checks:
            assert not np.isnan(e).any(), (tag, n, c, h, w, m)
            err = H.rel_err(e, o)
            assert err < 2e-5, (tag, (n, c, h, w, m, p, cm), err)
    print("f8-ok")


if __name__ == "__main__":
    main(sys.argv[1])
