"""__graft_entry__.smoke(): one small hot-path invocation on cuda:0 through the C-ABI, checked
against the CPU oracle (test infrastructure: the only place besides tests/ and bench.py's
cpu_baseline leg that touches oracle/)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run():
    import torch
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    import abi_harness as H
    from swapping_autoencoder_pytorch_amd import hip_lib as L
    hip = L.get()
    so = os.path.join(ROOT, "oracle", "libsae_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    ora = L.SaeLibrary(so, prefix="oracle_", device_only=False)
    rng = np.random.default_rng(0)
    dev = "cuda:0"
    # D-style downsampling slice: blur -> 3x3 stride-2 conv -> bias + leaky-ReLU, forward and backward
    x = rng.standard_normal((2, 16, 16, 16)).astype(np.float32)
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64.0
    w = rng.standard_normal((24, 16, 3, 3)).astype(np.float32)
    b = rng.standard_normal(24).astype(np.float32)
    outs = []
    for lib, d in ((hip, dev), (ora, None)):
        xb = H.upfirdn2d(lib, x.reshape(32, 16, 16, 1), k, pad=(2, 2, 2, 2), device=d).reshape(2, 16, 17, 17)
        desc = H.conv_desc(2, 16, 17, 17, 24, 3, 2, 0)
        y = H.conv(lib, 0, desc, xb, w, (2, 24, 8, 8), alpha=1.0 / 12.0, device=d)
        a = H.bias_act(lib, y, b, None, device=d)
        gy = np.ones_like(a)
        ga, gb = H.bias_act_bwd(lib, gy, a, device=d)
        gxb = H.conv(lib, 1, desc, ga, w, xb.shape, alpha=1.0 / 12.0, device=d)
        gw = H.conv(lib, 2, desc, xb, ga, w.shape, alpha=1.0 / 12.0, device=d)
        outs.append((a, gb, gxb, gw))
    for name, got, want in zip(("act", "grad_bias", "grad_input", "grad_weight"), outs[0], outs[1]):
        err = H.rel_err(got, want)
        assert err < 1e-4, (name, err)
    # the step's dominant kernels: a 3x3 stride-1 layer on the one-kernel Winograd route (forward with bias + leaky-ReLU, data
    # gradient, weight gradient) against the oracle's DIRECT convolution
    x1 = rng.standard_normal((2, 24, 16, 16)).astype(np.float32)
    w1 = rng.standard_normal((40, 24, 3, 3)).astype(np.float32)
    b1 = rng.standard_normal(40).astype(np.float32)
    g1 = rng.standard_normal((2, 40, 16, 16)).astype(np.float32)
    d1 = H.conv_desc(2, 24, 16, 16, 40, 3, 1, 1)
    al = 1.0 / (24 * 9) ** 0.5
    checks = (("winograd forward + bias + lrelu", H.wino_fused_conv(hip, x1, w1, alpha=al, bias=b1, act=(0.2, 2 ** 0.5), device=dev),
               H.conv_bias_act(ora, d1, x1, w1, b1, alpha=al)),
              ("winograd data gradient", H.wino_fused_conv(hip, g1, w1, alpha=al, transpose=True, device=dev),
               H.conv(ora, 1, d1, g1, w1, x1.shape, alpha=al)),
              ("winograd weight gradient", H.wino_fused_wgrad(hip, x1, g1, alpha=al, device=dev),
               H.conv(ora, 2, d1, x1, g1, w1.shape, alpha=al)))
    for name, got, want in checks:
        err = H.rel_err(got, want)
        assert err < 1e-4, (name, err)
    print("smoke ok: blur -> conv3x3/s2 -> bias+lrelu fwd/bwd and the one-kernel Winograd 3x3/s1 fwd / dgrad / wgrad on cuda:0 within "
          "1e-4 of the oracle")
