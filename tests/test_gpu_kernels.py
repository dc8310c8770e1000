"""GPU parity: the real gfx950 library, called through the C-ABI with device pointers, versus the
CPU oracle on the same seeded inputs.  fp32 tolerance 2e-5 relative to the output's max
magnitude per op (north star: <= 1e-4 relative per op)."""
import numpy as np
import pytest

import abi_harness as H
from kernel_cases import CONV_BX, BIAS_ACT_SHAPES, CONV_GPU, GEMM_CASES, K1_EPILOGUE, UPFIRDN_SMALL

pytestmark = pytest.mark.gpu
TOL = 2e-5
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip_lib():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from swapping_autoencoder_pytorch_amd import hip_lib as L
    return L.get()


UPFIRDN_GPU = UPFIRDN_SMALL + [
    ((64, 256, 256, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),
    ((64, 257, 257, 1), (4, 4), (1, 1), (1, 1), (1, 1, 1, 1)),
    ((32, 259, 259, 1), (3, 3), (1, 1), (1, 1), (0, 0, 0, 0)),
    ((3000, 8, 8, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),
    ((8, 64, 64, 1), (4, 4), (2, 2), (1, 1), (2, 1, 2, 1)),
    ((8, 64, 64, 1), (4, 4), (1, 1), (2, 2), (1, 1, 1, 1)),
]


@pytest.mark.parametrize("case", UPFIRDN_GPU, ids=lambda c: "x".join(map(str, c[0])) + "_k%dx%d" % c[1])
def test_upfirdn2d(hip_lib, oracle_lib, case):
    xs, ks, up, down, pad = case
    rng = np.random.default_rng(1)
    x = rng.standard_normal(xs).astype(np.float32)
    k = rng.standard_normal(ks).astype(np.float32)
    a = H.upfirdn2d(hip_lib, x, k, up, down, pad, device=DEV)
    o = H.upfirdn2d(oracle_lib, x, k, up, down, pad)
    assert a.shape == o.shape and not np.isnan(a).any()
    assert H.rel_err(a, o) < TOL


@pytest.mark.parametrize("case", [(4, 128, 64, 64, 256, 1, 1, 0), (2, 512, 16, 16, 512, 1, 1, 0), (8, 32, 64, 64, 64, 1, 1, 0),
                                  (2, 128, 33, 33, 128, 1, 2, 0)], ids=str)
def test_conv2d_residual(hip_lib, oracle_lib, case):
    from test_emu_kernels import CONV_RESIDUAL, conv_residual_case
    conv_residual_case(hip_lib, oracle_lib, case, device=DEV)
    for small in CONV_RESIDUAL[:3]:
        conv_residual_case(hip_lib, oracle_lib, small, device=DEV)


@pytest.mark.parametrize("case", K1_EPILOGUE + [(2, 16, 129, 129, 4, 1, (1, 1, 1, 1)), (2, 8, 64, 64, 4, 2, (2, 1, 2, 1)),
                                                  (300, 8, 5, 5, 4, 1, (1, 1, 1, 1))], ids=str)
def test_upfirdn2d_epilogue(hip_lib, oracle_lib, case):
    from test_emu_kernels import k1_epilogue_case
    k1_epilogue_case(hip_lib, oracle_lib, case, device=DEV)


@pytest.mark.parametrize("shape", BIAS_ACT_SHAPES + [(16, 128, 64, 64), (128, 32, 32, 32), (16, 512), (4, 7, 129, 129)], ids=str)
def test_bias_act(hip_lib, oracle_lib, shape):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape[1]).astype(np.float32)
    ref = rng.standard_normal(shape).astype(np.float32)
    for act, grad in [(3, 0), (3, 1), (3, 2), (1, 0)]:
        r = ref if grad else None
        a = H.bias_act(hip_lib, x, b, r, act, grad, device=DEV)
        o = H.bias_act(oracle_lib, x, b, r, act, grad)
        # one add, one select/mul, one mul per element: bit exact unless the compiler contracts
        # the (x + b) * alpha pair differently -> allow 1 ulp
        assert np.allclose(a, o, rtol=2e-7, atol=0), (act, grad)
    gx_a, gb_a = H.bias_act_bwd(hip_lib, x, ref, device=DEV)
    gx_o, gb_o = H.bias_act_bwd(oracle_lib, x, ref)
    assert np.allclose(gx_a, gx_o, rtol=2e-7, atol=0)
    scale = np.abs(gx_o).sum() / gb_o.size
    assert np.abs(gb_a - gb_o).max() <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize("case", CONV_GPU, ids=lambda c: "n%d_c%d_%dx%d_m%d_k%d_s%d_p%d_%s" % c)
def test_conv2d(hip_lib, oracle_lib, case):
    n, c, h, w, m, k, s, p, cm = case
    d = H.conv_desc(n, c, h, w, m, k, s, p, cm)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((c, m, k, k) if cm else (m, c, k, k)).astype(np.float32)
    gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
    for op, (a, b, shape) in enumerate([(x, wt, gy.shape), (gy, wt, x.shape), (x, gy, wt.shape)]):
        e = H.conv(hip_lib, op, d, a, b, shape, alpha=0.37, device=DEV)
        o = H.conv(oracle_lib, op, d, a, b, shape, alpha=0.37)
        assert not np.isnan(e).any(), op
        assert H.rel_err(e, o) < TOL, (op, H.rel_err(e, o))


CONV_BX_GPU = CONV_BX + [(2, 128, 32, 32, 128, 3, 1, 1, False), (8, 96, 4, 4, 96, 3, 1, 1, False),
                         (1, 256, 64, 64, 128, 3, 1, 1, True), (2, 72, 34, 34, 72, 3, 1, 0, False)]


@pytest.mark.parametrize("case", CONV_BX_GPU, ids=lambda c: "n%d_c%d_%dx%d_m%d_k%d_s%d_p%d_%s" % c)
def test_conv2d_bf16x6(hip_lib, oracle_lib, case):
    """SAE_CONV_MATH_BF16X6 (three-way bf16 split on the bf16 matrix cores) against the double
    oracle: must be in the fp32 error class, i.e. inside the tolerance of the exact-fp32 kernels."""
    n, c, h, w, m, k, s, p, cm = case
    d = H.conv_desc(n, c, h, w, m, k, s, p, cm)
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((n, c, h, w)) * np.exp(rng.uniform(-3, 3, (n, c, 1, 1)))).astype(np.float32)
    wt = rng.standard_normal((c, m, k, k) if cm else (m, c, k, k)).astype(np.float32)
    gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    ops = [(x, wt, gy.shape), (gy, wt, x.shape), (x, gy, wt.shape)]
    exact = [H.conv(hip_lib, op, d, a, bb, shape, alpha=0.37, device=DEV) for op, (a, bb, shape) in enumerate(ops)]
    hip_lib.call("set_conv_math", 1)
    try:
        for op, (a, bb, shape) in enumerate(ops):
            e = H.conv(hip_lib, op, d, a, bb, shape, alpha=0.37, device=DEV)
            o = H.conv(oracle_lib, op, d, a, bb, shape, alpha=0.37)
            assert not np.isnan(e).any(), op
            err, err_exact = H.rel_err(e, o), H.rel_err(exact[op], o)
            assert err < 3e-6, (op, err)
            assert err < 3 * err_exact + 2e-7, (op, err, err_exact)     # no worse than the fp32 MFMA chain
        e = H.conv_bias_act(hip_lib, d, x, wt, b, alpha=0.11, device=DEV)
        o = H.conv_bias_act(oracle_lib, d, x, wt, b, alpha=0.11)
        assert H.rel_err(e, o) < 3e-6
    finally:
        hip_lib.call("set_conv_math", 0)


def _l1_scale(oracle_lib, op, d, a, b, shape):
    """sum_k |a_k b_k| per output element (the condition scale of the contraction), via the oracle on |a|, |b|."""
    return H.conv(oracle_lib, op, d, np.abs(a), np.abs(b), shape, alpha=1.0).astype(np.float64)


@pytest.mark.parametrize("kind", ["scales_1e30", "cancellation", "tiny_operands", "huge_dynamic_range_within_channel"])
def test_conv2d_bf16x6_adversarial(hip_lib, oracle_lib, kind):
    """Adversarial operand ranges for the bf16x6 split arithmetic (promotion condition ii).

    Error is measured against the double oracle relative to the condition scale L1 = sum_k |a_k b_k| of each
    output (the only scale any fp32 contraction can be held to), for BOTH arithmetics:
      * scales_1e30: per-channel magnitudes 1e-30 .. 1e+30 on x, compensated on w (outputs O(1)): every split piece
        stays a normal number, the bound is the nominal one;
      * cancellation: channel pairs that cancel to ~1e-6 of their magnitude: the error relative to L1 must not grow;
      * huge_dynamic_range_within_channel: magnitudes 1e-12 .. 1e+12 inside one plane;
      * tiny_operands: |x| ~ 1e-34 (below 2^-108 the third bf16 piece is subnormal and the matrix core flushes it;
        below 2^-117 the second one too): the documented limit of the split arithmetic — relative accuracy
        degrades from 2^-26 towards 2^-17 there, and the test pins exactly that bound (include/sae_hip.h)."""
    n, c, h, w, m, k, s, p = 2, 64, 16, 16, 96, 3, 1, 1
    d = H.conv_desc(n, c, h, w, m, k, s, p)
    rng = np.random.default_rng(101)
    x = rng.standard_normal((n, c, h, w))
    wt = rng.standard_normal((m, c, k, k))
    gy = rng.standard_normal((n, m, d.oh, d.ow))
    bound = 1e-6          # the exact-fp32 chain itself reaches ~4e-7 of L1 on these inputs
    if kind == "scales_1e30":
        e = rng.uniform(-30, 30, c)
        x *= (10.0 ** e)[None, :, None, None]
        wt *= (10.0 ** -e)[None, :, None, None]
    elif kind == "cancellation":
        x[:, 1::2] = -x[:, 0::2] * (1.0 + 1e-6 * rng.standard_normal(x[:, 0::2].shape))
        wt[:, 1::2] = wt[:, 0::2]
    elif kind == "huge_dynamic_range_within_channel":
        x *= 10.0 ** rng.uniform(-12, 12, x.shape)
        gy *= 10.0 ** rng.uniform(-12, 12, gy.shape)
    elif kind == "tiny_operands":
        x *= 1e-34
        wt *= 1e3
        gy *= 1e-34
        bound = 2.0 ** -16
    x, wt, gy = x.astype(np.float32), wt.astype(np.float32), gy.astype(np.float32)
    ops = [(x, wt, gy.shape), (gy, wt, x.shape), (x, gy, wt.shape)]
    rows = []
    for op, (a, b, shape) in enumerate(ops):
        o = H.conv(oracle_lib, op, d, a, b, shape, alpha=1.0).astype(np.float64)
        l1 = np.maximum(_l1_scale(oracle_lib, op, d, a, b, shape), 1e-300)
        exact = H.conv(hip_lib, op, d, a, b, shape, alpha=1.0, device=DEV).astype(np.float64)
        hip_lib.call("set_conv_math", 1)
        try:
            split = H.conv(hip_lib, op, d, a, b, shape, alpha=1.0, device=DEV).astype(np.float64)
        finally:
            hip_lib.call("set_conv_math", 0)
        assert np.isfinite(split).all(), (kind, op)
        e_exact = float((np.abs(exact - o) / l1).max())
        e_split = float((np.abs(split - o) / l1).max())
        rows.append((op, e_exact, e_split))
        assert e_split <= bound, (kind, op, e_exact, e_split)
        if kind != "tiny_operands":
            assert e_split <= 3 * e_exact + 2e-7, (kind, op, e_exact, e_split)
    print("bf16x6 adversarial", kind, rows)


@pytest.mark.parametrize("mnk", GEMM_CASES + [(128, 2048, 3072), (16, 512, 2048)], ids=str)
def test_gemm(hip_lib, oracle_lib, mnk):
    m, n, k = mnk
    rng = np.random.default_rng(4)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = rng.standard_normal((n, k)).astype(np.float32)
    gy = rng.standard_normal((m, n)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    for args in [(x, w, bias, m, n, k, k, 1, 1, k, 0.5), (gy, w, None, m, k, n, n, 1, k, 1, 1.0),
                 (gy, x, None, n, k, m, 1, n, k, 1, 1.0)]:
        e = H.gemm(hip_lib, *args, device=DEV)
        o = H.gemm(oracle_lib, *args)
        assert H.rel_err(e, o) < TOL
        # the form the product calls: K split across workgroups where sae_gemm_workspace says so (skinny shapes)
        es, n_ws = H.gemm(hip_lib, *args, device=DEV, split=True)
        assert not np.isnan(es).any() and H.rel_err(es, o) < TOL, n_ws


@pytest.mark.parametrize("shape", [(3, 1, 1), (2, 5, 7), (4, 16, 16), (1, 33, 20)], ids=str)
def test_upsample2x(hip_lib, oracle_lib, shape):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape).astype(np.float32)
    res = rng.standard_normal((shape[0], 2 * shape[1], 2 * shape[2])).astype(np.float32)
    for r in (res, None):
        a = H.upsample2x_add(hip_lib, x, r, 0.7, device=DEV)
        o = H.upsample2x_add(oracle_lib, x, r, 0.7)
        assert H.rel_err(a, o) < 1e-6
    a = H.upsample2x_bwd(hip_lib, res, 0.7, device=DEV)
    o = H.upsample2x_bwd(oracle_lib, res, 0.7)
    assert H.rel_err(a, o) < 1e-6
    # adjointness: <up(x), g> == <x, up^T(g)>
    up = H.upsample2x_add(oracle_lib, x, None, 1.0).astype(np.float64)
    dn = H.upsample2x_bwd(oracle_lib, res, 1.0).astype(np.float64)
    assert abs((up * res).sum() - (x * dn).sum()) < 1e-4 * max(1.0, abs((up * res).sum()))


@pytest.mark.parametrize("case", [(2, 8, 8, 8, 32, 3, 1, 1), (1, 5, 17, 17, 40, 3, 2, 0), (2, 64, 8, 8, 40, 3, 1, 1),
                                  (2, 33, 7, 7, 130, 1, 1, 0)], ids=str)
def test_conv2d_bias_act(hip_lib, oracle_lib, case):
    n, c, h, w, m, k, s, p = case
    d = H.conv_desc(n, c, h, w, m, k, s, p)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((m, c, k, k)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    for bias in (b, None):
        e = H.conv_bias_act(hip_lib, d, x, wt, bias, alpha=0.11, device=DEV)
        o = H.conv_bias_act(oracle_lib, d, x, wt, bias, alpha=0.11)
        assert H.rel_err(e, o) < TOL


@pytest.mark.parametrize("shape", [(2, 5, 16, 16), (3, 8, 4, 8), (1, 70, 32, 36), (2, 600, 8, 8), (4, 128, 64, 64),
                                   (2, 32, 256, 256)], ids=str)
def test_noise_bias_act(hip_lib, oracle_lib, shape):
    """StyledConv glue kernels (csrc/modulate.hip) against the double-accumulating oracle."""
    rng = np.random.default_rng(21)
    n, c = shape[:2]
    x = rng.standard_normal(shape).astype(np.float32)
    noise = rng.standard_normal((n, 1) + shape[2:]).astype(np.float32)
    nw = np.array([0.37], np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    for nz, bias in ((noise, b), (None, b), (noise, None)):
        e = H.noise_bias_act(hip_lib, x, nz, nw, bias, device=DEV)
        o = H.noise_bias_act(oracle_lib, x, nz, nw, bias)
        assert H.rel_err(e, o) < 1e-6
    y = H.noise_bias_act(oracle_lib, x, noise, nw, b)
    gy = rng.standard_normal(shape).astype(np.float32)
    gx_e, gb_e, gw_e = H.noise_bias_act_bwd(hip_lib, gy, y, noise, device=DEV)
    gx_o, gb_o, gw_o = H.noise_bias_act_bwd(oracle_lib, gy, y, noise)
    assert np.array_equal(gx_e, gx_o)
    scale = float(np.abs(gx_o).sum() / c)
    assert np.abs(gb_e - gb_o).max() <= 1e-6 * scale
    assert abs(float(gw_e[0]) - float(gw_o[0])) <= 1e-6 * float(np.abs(gx_o).sum())
    s = rng.standard_normal((n, c)).astype(np.float32)
    g2_e, gs_e = H.plane_scale_dot(hip_lib, gy, x, s, device=DEV)
    g2_o, gs_o = H.plane_scale_dot(oracle_lib, gy, x, s)
    assert np.array_equal(g2_e, g2_o)
    assert np.abs(gs_e - gs_o).max() <= 1e-6 * float(np.abs(gy * x).sum() / (n * c))
    for nz in (noise, None):      # the fused pair: plane_scale_dot + the producer's noise / bias / activation backward
        fx, fs, fb, fw = H.plane_scale_dot_act(hip_lib, gy, y, s, nz, device=DEV)
        ox, os_, ob, ow_ = H.plane_scale_dot_act(oracle_lib, gy, y, s, nz)
        assert np.array_equal(fx, ox)
        assert np.abs(fs - os_).max() <= 1e-6 * float(np.abs(gy * y).sum() / (n * c))
        assert np.abs(fb - ob).max() <= 1e-6 * float(np.abs(ox).sum() / c)
        assert nz is None or abs(float(fw[0]) - float(ow_[0])) <= 1e-6 * float(np.abs(ox).sum())


@pytest.mark.parametrize("shape", [(5, 3, 3, 3), (512, 512, 3, 3), (128, 256, 3, 3), (3, 128, 1, 1)], ids=str)
def test_weight_demod(hip_lib, oracle_lib, shape):
    """sae_weight_demod_f32 / _bwd_f32 against the oracle (pinned to the reference's ATen sequence in the CPU suite)."""
    rng = np.random.default_rng(17)
    w = rng.standard_normal(shape).astype(np.float32)
    geff = rng.standard_normal(shape).astype(np.float32)
    alpha = float(1.0 / np.sqrt(np.prod(shape[1:])))
    d_o = H.weight_demod(oracle_lib, w, alpha)
    assert np.allclose(H.weight_demod(hip_lib, w, alpha, device=DEV), d_o, rtol=3e-7, atol=0)
    gw_o = H.weight_demod_bwd(oracle_lib, geff, w, d_o, alpha)
    gw_e = H.weight_demod_bwd(hip_lib, geff, w, d_o, alpha, device=DEV)
    assert np.allclose(gw_e, gw_o, rtol=1e-5, atol=1e-6 * float(np.abs(gw_o).max()))


@pytest.mark.parametrize("case", [(2, 3, 32, 32, 4, 16, 0.125, 0.25), (1, 2, 20, 28, 3, 9, 0.3, 1.0),
                                  (4, 3, 256, 256, 8, 128, 0.125, 0.25)], ids=str)
def test_random_crop(hip_lib, oracle_lib, case):
    """Patch sampler kernels against the oracle (itself pinned to F.grid_sample in the CPU suite)."""
    n, c, h, w, crops, size, lo, hi = case
    rng = np.random.default_rng(31)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    k = n * crops
    flip = np.round(rng.random(k)) * 2 - 1
    scale = rng.random((k, 2)) * (hi - lo) + lo
    offset = (rng.random((k, 2)) * 2 - 1) * (1 - scale)
    params = np.stack([flip, scale[:, 0], scale[:, 1], offset[:, 0], offset[:, 1]], 1).astype(np.float32)
    gy = rng.standard_normal((k, c, size, size)).astype(np.float32)
    e = H.random_crop(hip_lib, x, params, size, crops, device=DEV)
    o = H.random_crop(oracle_lib, x, params, size, crops)
    assert H.rel_err(e, o) < 5e-6
    eb = H.random_crop_bwd(hip_lib, gy, params, x.shape, crops, device=DEV)
    ob = H.random_crop_bwd(oracle_lib, gy, params, x.shape, crops)
    assert H.rel_err(eb, ob) < 5e-6


@pytest.mark.parametrize("case", [((2, 3, 8, 8), (1, 1, 1, 1)), ((1, 2, 5, 7), (2, 1, 0, 3)), ((16, 32, 256, 256), (1, 1, 1, 1)),
                                  ((4, 64, 128, 128), (1, 2, 1, 2))], ids=str)
def test_reflect_pad(hip_lib, oracle_lib, case):
    shape, pads = case
    rng = np.random.default_rng(41)
    x = rng.standard_normal(shape).astype(np.float32)
    o = H.reflect_pad(oracle_lib, x, pads)
    assert np.array_equal(H.reflect_pad(hip_lib, x, pads, device=DEV), o)
    gy = rng.standard_normal(o.shape).astype(np.float32)
    assert np.allclose(H.reflect_pad_adj(hip_lib, gy, pads, device=DEV), H.reflect_pad_adj(oracle_lib, gy, pads),
                       rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", [((512, 256, 256, 1), (4, 4), (2, 2, 2, 2)), ((256, 257, 257, 1), (4, 4), (1, 1, 1, 1)),
                                  ((1024, 64, 64, 1), (4, 4), (2, 2, 2, 2)), ((64, 129, 129, 1), (4, 4), (1, 1, 1, 1)),
                                  ((96, 259, 259, 1), (3, 3), (0, 0, 0, 0)), ((3, 67, 70, 1), (4, 4), (2, 2, 2, 2))],
                         ids=lambda c: "x".join(map(str, c[0])) + "_k%dx%d" % c[1])
def test_streaming_blur_is_the_strip_blur_bit_for_bit(oracle_lib, case, monkeypatch):
    """blur_stream_kernel (csrc/upfirdn2d.hip) at the step's plane sizes: same bits as the LDS-strip kernels (the tuning build of
    the same sources has the dispatch knob), oracle parity on a slice of planes.  upfirdn2d_kernel.cu:52-137 of the reference."""
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    from tuning import build_tuning
    from test_emu_kernels import k1_stream_case
    lib = SaeLibrary(build_tuning.build())
    xs, ks, pad = case
    small = ((min(xs[0], 4),) + xs[1:], ks, pad)        # the oracle walks a few planes; bit-identity is checked on all of them below
    k1_stream_case(lib, oracle_lib, small, monkeypatch, device=DEV)
    rng = np.random.default_rng(9)
    x = rng.standard_normal(xs).astype(np.float32)
    k = rng.standard_normal(ks).astype(np.float32)
    monkeypatch.setenv("SAE_K1_STREAM", "2")
    a = H.upfirdn2d(lib, x, k, (1, 1), (1, 1), pad, device=DEV)
    monkeypatch.setenv("SAE_K1_STREAM", "0")
    b = H.upfirdn2d(lib, x, k, (1, 1), (1, 1), pad, device=DEV)
    assert not np.isnan(a).any() and np.array_equal(a, b)
