"""Style-modulated convolution as one C-ABI call per operation (include/sae_hip.h: sae_modconv2d_*): the oracle
restates ModulatedConv2d literally (stylegan2_layers.py:280-321), the emulated kernels (CPU) and the real kernels
(``-m gpu``) stage the factors on the fly and must agree with it."""
import ctypes as C

import numpy as np
import pytest

import abi_harness as H
from modconv_cases import MODCONV_CASES, run_case


def test_oracle_modconv_is_scale_then_conv(oracle_lib):
    """the oracle's modulated forward == its plain forward on explicitly scaled operands (the reference's two steps)"""
    n, c, h, w, m, k = 2, 6, 7, 7, 10, 3
    d = H.conv_desc(n, c, h, w, m, k, 1, 1)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((m, c, k, k)).astype(np.float32)
    s = rng.standard_normal((n, c)).astype(np.float32)
    dm = rng.uniform(0.5, 2, m).astype(np.float32)
    a = H.modconv(oracle_lib, 0, d, x, wt, (n, m, h, w), x_scale=s, wm_scale=dm, alpha=0.5)
    b = H.conv(oracle_lib, 0, d, x * s[:, :, None, None], wt * dm[:, None, None, None], (n, m, h, w), alpha=0.5)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("case", MODCONV_CASES, ids=lambda c: "n%d_c%d_%dx%d_m%d_k%d_s%d_p%d_%s" % c)
def test_emulated_kernels_vs_oracle(emu_lib, oracle_lib, case):
    run_case(emu_lib, oracle_lib, case)


def test_meaningless_factors_are_refused(emu_lib, oracle_lib):
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeError
    d = H.conv_desc(1, 4, 6, 6, 8, 3, 1, 1)
    x = np.zeros((1, 4, 6, 6), np.float32)
    wt = np.zeros((8, 4, 3, 3), np.float32)
    gy = np.zeros((1, 8, 6, 6), np.float32)
    one = np.ones((1, 8), np.float32)
    for lib in (emu_lib, oracle_lib):
        with pytest.raises(SaeError):
            H.modconv(lib, 0, d, x, wt, gy.shape, y_scale=one)
        with pytest.raises(SaeError):
            H.modconv(lib, 1, d, gy, wt, x.shape, x_scale=np.ones((1, 4), np.float32))
        with pytest.raises(SaeError):
            H.modconv(lib, 2, d, x, gy, wt.shape, wm_scale=np.ones(8, np.float32))


MODCONV_GPU = MODCONV_CASES + [
    (4, 128, 64, 64, 128, 3, 1, 1, False),     # G block at reduced batch: one image per tile, 128x128 tile
    (2, 128, 65, 65, 256, 3, 2, 0, True),      # G upsampling block 256 -> 128, 32 -> 65
    (2, 128, 64, 64, 3, 1, 1, 0, False),       # ToRGB
    (16, 512, 16, 16, 512, 3, 1, 1, False),    # 16x16 tail: split-K, one image per tile
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", MODCONV_GPU, ids=lambda c: "n%d_c%d_%dx%d_m%d_k%d_s%d_p%d_%s" % c)
def test_gpu_kernels_vs_oracle(oracle_lib, case):
    from swapping_autoencoder_pytorch_amd import hip_lib
    run_case(hip_lib.get(), oracle_lib, case, device="cuda:0")


@pytest.mark.gpu
def test_gpu_bf16x6_refuses_activation_factors_but_takes_weight_factors(oracle_lib):
    from swapping_autoencoder_pytorch_amd import hip_lib
    lib = hip_lib.get()
    n, c, h, w, m, k = 2, 64, 8, 8, 70, 3
    d = H.conv_desc(n, c, h, w, m, k, 1, 1)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((m, c, k, k)).astype(np.float32)
    dm = rng.uniform(0.5, 2, m).astype(np.float32)
    lib.call("set_conv_math", 1)
    try:
        with pytest.raises(hip_lib.SaeError):
            H.modconv(lib, 0, d, x, wt, (n, m, h, w), x_scale=np.ones((n, c), np.float32), device="cuda:0")
        e = H.modconv(lib, 0, d, x, wt, (n, m, h, w), wm_scale=dm, alpha=0.3, device="cuda:0")
        o = H.modconv(oracle_lib, 0, d, x, wt, (n, m, h, w), wm_scale=dm, alpha=0.3)
        assert H.rel_err(e, o) < 3e-6
    finally:
        lib.call("set_conv_math", 0)
