"""Parity of a csrc/winograd_fused.hip variant library (tools/build_wf_variant.sh) on the GPU: the one-kernel route's test cases against
the oracle, and two layers of the step's size against the product's direct kernels.   python tools/wf_variant_check.py wf_pk"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi_harness as H  # noqa: E402
import test_winograd as T  # noqa: E402
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402

name = sys.argv[1]
var = hip_lib.SaeLibrary(os.path.join(ROOT, "tools", "variants", name + ".so"))
ora = hip_lib.SaeLibrary(os.path.join(ROOT, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
T._fused_route(var, ora, "cuda:0")
T._fused_wgrad(var, ora, "cuda:0")
prod = hip_lib.get()
rng = np.random.default_rng(43)
worst = 0.0
for n, c, m, side in [(3, 128, 128, 64), (2, 200, 72, 32), (2, 512, 512, 64), (1, 64, 64, 256)]:
    x = rng.standard_normal((n, c, side, side)).astype(np.float32)
    wt = (rng.standard_normal((m, c, 3, 3)) / (3 * c ** 0.5)).astype(np.float32)
    gy = rng.standard_normal((n, m, side, side)).astype(np.float32)
    d = H.conv_desc(n, c, side, side, m, 3, 1, 1)
    e1 = H.rel_err(H.wino_fused_conv(var, x, wt, device="cuda:0"), H.conv(prod, 0, d, x, wt, gy.shape, alpha=1.0, device="cuda:0"))
    e2 = H.rel_err(H.wino_fused_conv(var, gy, wt, transpose=True, device="cuda:0"), H.conv(prod, 1, d, gy, wt, x.shape, alpha=1.0, device="cuda:0"))
    worst = max(worst, e1, e2)
    assert e1 < 2e-5 and e2 < 2e-5, (n, c, m, side, e1, e2)
print("%s: parity ok (oracle cases + four step-size layers against the direct kernels, worst %.2e)" % (name, worst))
