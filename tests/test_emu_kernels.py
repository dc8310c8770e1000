"""CPU: the product's kernel sources, compiled for the host against the hipemu header
(tests/emu), versus the CPU oracle.  Exercises the real tile/index arithmetic, LDS staging,
barriers and MFMA lane layouts without a GPU."""
import numpy as np
import pytest

import abi_harness as H
from kernel_cases import BIAS_ACT_SHAPES, CONV_BX, CONV_SMALL, GEMM_CASES, K1_EPILOGUE, UPFIRDN_SMALL

TOL = 2e-5   # fp32 kernel vs double-accumulating oracle, relative to the output's max magnitude


@pytest.mark.parametrize("case", UPFIRDN_SMALL, ids=lambda c: "x".join(map(str, c[0])) + "_k%dx%d" % c[1])
def test_upfirdn2d(emu_lib, oracle_lib, case):
    xs, ks, up, down, pad = case
    rng = np.random.default_rng(1)
    x = rng.standard_normal(xs).astype(np.float32)
    k = rng.standard_normal(ks).astype(np.float32)
    a = H.upfirdn2d(emu_lib, x, k, up, down, pad)
    o = H.upfirdn2d(oracle_lib, x, k, up, down, pad)
    assert a.shape == o.shape
    assert not np.isnan(a).any()
    assert H.rel_err(a, o) < TOL


# planes the streaming blur takes (csrc/upfirdn2d.hip: blur_streams -- at least 32 wide, 16 output rows, 2 ... 4 taps): 2^k and
# 2^k + 1 wide rows, the left-border window shift for every pad_x0, strips of unequal height, several planes per wave
K1_STREAM = [((3, 67, 70, 1), (4, 4), (2, 2, 2, 2)), ((3, 33, 33, 1), (4, 4), (1, 1, 1, 1)), ((2, 64, 64, 1), (4, 4), (2, 2, 2, 2)),
             ((2, 65, 65, 1), (4, 4), (1, 1, 1, 1)), ((5, 40, 32, 1), (3, 3), (0, 0, 0, 0)), ((1, 300, 64, 1), (4, 4), (2, 2, 1, 1)),
             ((7, 16, 36, 1), (4, 4), (3, 0, 3, 0)), ((2, 129, 129, 1), (4, 4), (1, 1, 1, 1)), ((3, 48, 35, 1), (2, 2), (1, 0, 1, 0))]


def k1_stream_case(lib, oracle_lib, case, monkeypatch, device=None):
    """blur_stream_kernel against the oracle and -- bit for bit -- against the LDS-strip kernels it replaces on these planes (the
    dispatch knob SAE_K1_STREAM exists in tuning builds: the emulator, tests/tuning)."""
    xs, ks, pad = case
    rng = np.random.default_rng(3)
    x = rng.standard_normal(xs).astype(np.float32)
    k = rng.standard_normal(ks).astype(np.float32)
    monkeypatch.setenv("SAE_K1_STREAM", "2")
    a = H.upfirdn2d(lib, x, k, (1, 1), (1, 1), pad, device=device)
    monkeypatch.setenv("SAE_K1_STREAM", "0")
    b = H.upfirdn2d(lib, x, k, (1, 1), (1, 1), pad, device=device)
    o = H.upfirdn2d(oracle_lib, x, k, (1, 1), (1, 1), pad)
    assert not np.isnan(a).any() and H.rel_err(a, o) < 2e-6
    assert np.array_equal(a, b), float(np.abs(a - b).max())
    # ... and with the forward epilogue of StyledConv's upsampling form
    ch = 1 if xs[0] % 2 else 2
    outer = xs[0] // ch
    oh, ow = a.shape[1], a.shape[2]
    nz = rng.standard_normal((outer, oh, ow)).astype(np.float32)
    nw = np.array([0.7], np.float32)
    bias = rng.standard_normal(ch).astype(np.float32)
    monkeypatch.setenv("SAE_K1_STREAM", "2")
    ya = H.upfirdn2d_noise_bias_act(lib, x[..., 0], k, pad, nz, nw, bias, ch, device=device)
    monkeypatch.setenv("SAE_K1_STREAM", "0")
    yb = H.upfirdn2d_noise_bias_act(lib, x[..., 0], k, pad, nz, nw, bias, ch, device=device)
    yo = H.upfirdn2d_noise_bias_act(oracle_lib, x[..., 0], k, pad, nz, nw, bias, ch)
    assert H.rel_err(ya, yo) < 2e-6 and np.array_equal(ya, yb)


@pytest.mark.parametrize("case", K1_STREAM, ids=lambda c: "x".join(map(str, c[0])) + "_k%dx%d" % c[1])
def test_streaming_blur_is_the_strip_blur_bit_for_bit(emu_lib, oracle_lib, case, monkeypatch):
    k1_stream_case(emu_lib, oracle_lib, case, monkeypatch)


def k1_epilogue_case(lib, oracle_lib, case, device=None):
    """Every epilogue combination of sae_upfirdn2d_epilogue_f32 against (a) the oracle's restatement and (b) the SAME
    library's separate calls (upfirdn2d, then the add, then the K2 backward): the fused value must be bit-identical to the
    unfused one -- the FIR sum is rounded to float before the elementwise work either way."""
    outer, ch, ih, iw, taps, up, pad = case
    rng = np.random.default_rng(23)
    x = rng.standard_normal((outer * ch, ih, iw)).astype(np.float32)
    k = rng.standard_normal((taps, taps)).astype(np.float32)
    plain = H.upfirdn2d(lib, x[..., None], k, (up, up), (1, 1), pad, device=device)[..., 0]
    old = rng.standard_normal(plain.shape).astype(np.float32)
    ref = rng.standard_normal(plain.shape).astype(np.float32)
    for acc in (False, True):
        for act in (False, True):
            kw = dict(act_ref=ref if act else None, channels=ch)
            y, gb = H.upfirdn2d_epilogue(lib, x, k, up, pad, y_old=old.copy() if acc else None, device=device, **kw)
            yo, gbo = H.upfirdn2d_epilogue(oracle_lib, x, k, up, pad, y_old=old.copy() if acc else None, **kw)
            assert not np.isnan(y).any()
            assert H.rel_err(y, yo) < TOL, (acc, act)
            want = plain + old if acc else plain
            if act:
                shaped = want.reshape(outer, ch, -1)
                gx, gbs = H.bias_act_bwd(lib, shaped, ref.reshape(outer, ch, -1), device=device)
                want = gx.reshape(want.shape)
                assert np.allclose(gb, gbo, rtol=0, atol=2e-5 * np.abs(want).sum() / ch + 1e-6)
                assert np.allclose(gb, gbs, rtol=0, atol=2e-5 * np.abs(want).sum() / ch + 1e-6)
            assert np.array_equal(y, want), (acc, act)
    if up == 1:
        # the forward activation on the way out (sae_upfirdn2d_noise_bias_act_f32): against the oracle's blur, then NoiseInjection +
        # FusedLeakyReLU, and against this library's own two calls (bit-identical where the two-call form exists: hw % 4 == 0)
        noise = rng.standard_normal((outer,) + plain.shape[1:]).astype(np.float32)
        nw = np.array([0.37], np.float32)
        bias = rng.standard_normal(ch).astype(np.float32)
        for nz, b_ in ((noise, bias), (None, bias), (noise, None)):
            y = H.upfirdn2d_noise_bias_act(lib, x, k, pad, nz, nw if nz is not None else None, b_, ch, device=device)
            yo = H.upfirdn2d_noise_bias_act(oracle_lib, x, k, pad, nz, nw if nz is not None else None, b_, ch)
            assert not np.isnan(y).any()
            assert H.rel_err(y, yo) < TOL, ("forward activation", nz is not None, b_ is not None)
            hw = plain.shape[1] * plain.shape[2]
            if hw % 4 == 0:
                two = H.noise_bias_act(lib, plain.reshape(outer, ch, plain.shape[1], plain.shape[2]),
                                       None if nz is None else nz.reshape(outer, 1, plain.shape[1], plain.shape[2]), nw, b_,
                                       device=device)
                assert H.rel_err(y.reshape(two.shape), two) < 2e-7


@pytest.mark.parametrize("case", K1_EPILOGUE, ids=str)
def test_upfirdn2d_epilogue(emu_lib, oracle_lib, case):
    k1_epilogue_case(emu_lib, oracle_lib, case)


@pytest.mark.parametrize("shape", BIAS_ACT_SHAPES, ids=str)
def test_bias_act(emu_lib, oracle_lib, shape):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape[1]).astype(np.float32)
    ref = rng.standard_normal(shape).astype(np.float32)
    for act, grad in [(3, 0), (3, 1), (3, 2), (1, 0), (1, 1)]:
        r = ref if grad else None
        a = H.bias_act(emu_lib, x, b, r, act, grad)
        o = H.bias_act(oracle_lib, x, b, r, act, grad)
        assert np.array_equal(a, o), (act, grad)      # elementwise fp32: bit exact
    a = H.bias_act(emu_lib, x, None, None)
    o = H.bias_act(oracle_lib, x, None, None)
    assert np.array_equal(a, o)
    gx_a, gb_a = H.bias_act_bwd(emu_lib, x, ref)
    gx_o, gb_o = H.bias_act_bwd(oracle_lib, x, ref)
    assert np.array_equal(gx_a, gx_o)
    assert np.allclose(gb_a, gb_o, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", CONV_SMALL, ids=lambda c: "n%d_c%d_%dx%d_m%d_k%d_s%d_p%d_%s" % c)
def test_conv2d(emu_lib, oracle_lib, case):
    n, c, h, w, m, k, s, p, cm = case
    d = H.conv_desc(n, c, h, w, m, k, s, p, cm)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((c, m, k, k) if cm else (m, c, k, k)).astype(np.float32)
    gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
    for op, (a, b, shape) in enumerate([(x, wt, gy.shape), (gy, wt, x.shape), (x, gy, wt.shape)]):
        e = H.conv(emu_lib, op, d, a, b, shape, alpha=0.37)
        o = H.conv(oracle_lib, op, d, a, b, shape, alpha=0.37)
        assert not np.isnan(e).any(), op
        assert H.rel_err(e, o) < TOL, (op, H.rel_err(e, o))


@pytest.mark.parametrize("case", CONV_BX, ids=lambda c: "n%d_c%d_%dx%d_m%d_k%d_s%d_p%d_%s" % c)
def test_conv2d_bf16x6(emu_lib, oracle_lib, case):
    """sae_set_conv_math(SAE_CONV_MATH_BF16X6): three-way bf16 split, six MFMAs per product block.
    Must stay in the fp32 error class (same tolerance as the exact-fp32 kernels)."""
    n, c, h, w, m, k, s, p, cm = case
    d = H.conv_desc(n, c, h, w, m, k, s, p, cm)
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((n, c, h, w)) * np.exp(rng.uniform(-3, 3, (n, c, 1, 1)))).astype(np.float32)
    wt = rng.standard_normal((c, m, k, k) if cm else (m, c, k, k)).astype(np.float32)
    gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    assert emu_lib.query("get_conv_math") == 0
    emu_lib.call("set_conv_math", 1)
    try:
        assert emu_lib.query("get_conv_math") == 1
        for op, (a, bb, shape) in enumerate([(x, wt, gy.shape), (gy, wt, x.shape), (x, gy, wt.shape)]):
            e = H.conv(emu_lib, op, d, a, bb, shape, alpha=0.37)
            o = H.conv(oracle_lib, op, d, a, bb, shape, alpha=0.37)
            assert not np.isnan(e).any(), op
            assert H.rel_err(e, o) < 3e-6, (op, H.rel_err(e, o))
        e = H.conv_bias_act(emu_lib, d, x, wt, b, alpha=0.11, device=None)
        o = H.conv_bias_act(oracle_lib, d, x, wt, b, alpha=0.11)
        assert H.rel_err(e, o) < 3e-6
    finally:
        emu_lib.call("set_conv_math", 0)
    with pytest.raises(Exception):
        emu_lib.call("set_conv_math", 7)


@pytest.mark.parametrize("mnk", GEMM_CASES, ids=str)
def test_gemm(emu_lib, oracle_lib, mnk):
    m, n, k = mnk
    rng = np.random.default_rng(4)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = rng.standard_normal((n, k)).astype(np.float32)
    gy = rng.standard_normal((m, n)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    for args in [(x, w, bias, m, n, k, k, 1, 1, k, 0.5),        # y = x W^T + b
                 (gy, w, None, m, k, n, n, 1, k, 1, 1.0),        # gx = gy W
                 (gy, x, None, n, k, m, 1, n, k, 1, 1.0)]:       # gW = gy^T x
        e = H.gemm(emu_lib, *args)
        o = H.gemm(oracle_lib, *args)
        assert H.rel_err(e, o) < TOL
        # the form the product calls: K split across workgroups where sae_gemm_workspace says so (skinny shapes)
        es, n_ws = H.gemm(emu_lib, *args, split=True)
        assert not np.isnan(es).any() and H.rel_err(es, o) < TOL, n_ws


def test_argument_errors(emu_lib):
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeError
    x = np.zeros((1, 4, 4, 1), np.float32)
    k = np.zeros((4, 4), np.float32)
    y = np.zeros((1, 4, 4, 1), np.float32)
    with pytest.raises(SaeError):   # up_x = 0
        emu_lib.call("upfirdn2d_f32", x.ctypes.data, k.ctypes.data, y.ctypes.data, 1, 4, 4, 1, 4, 4, 0, 1, 1, 1,
                     2, 1, 2, 1, None)
    with pytest.raises(SaeError):
        H.bias_act(emu_lib, x, None, None, act=2)
    d = H.conv_desc(1, 4, 8, 8, 4, 5, 1, 2)     # 5x5 taps: unsupported, must fail loudly
    with pytest.raises(SaeError):
        H.conv(emu_lib, 0, d, np.zeros((1, 4, 8, 8), np.float32), np.zeros((4, 4, 5, 5), np.float32), (1, 4, 8, 8))


@pytest.mark.parametrize("shape", [(3, 1, 1), (2, 5, 7), (4, 16, 16), (1, 33, 20), (2, 3, 130), (1, 5, 64)], ids=str)
def test_upsample2x(emu_lib, oracle_lib, shape):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape).astype(np.float32)
    res = rng.standard_normal((shape[0], 2 * shape[1], 2 * shape[2])).astype(np.float32)
    for r in (res, None):
        a = H.upsample2x_add(emu_lib, x, r, 0.7, device=None)
        o = H.upsample2x_add(oracle_lib, x, r, 0.7)
        assert H.rel_err(a, o) < 1e-6
    a = H.upsample2x_bwd(emu_lib, res, 0.7, device=None)
    o = H.upsample2x_bwd(oracle_lib, res, 0.7)
    assert H.rel_err(a, o) < 1e-6
    # adjointness: <up(x), g> == <x, up^T(g)>
    up = H.upsample2x_add(oracle_lib, x, None, 1.0).astype(np.float64)
    dn = H.upsample2x_bwd(oracle_lib, res, 1.0).astype(np.float64)
    assert abs((up * res).sum() - (x * dn).sum()) < 1e-4 * max(1.0, abs((up * res).sum()))


@pytest.mark.parametrize("case", [(2, 8, 8, 8, 32, 3, 1, 1), (1, 5, 17, 17, 40, 3, 2, 0), (2, 64, 8, 8, 40, 3, 1, 1),
                                  (2, 33, 7, 7, 130, 1, 1, 0)], ids=str)
def test_conv2d_bias_act(emu_lib, oracle_lib, case):
    n, c, h, w, m, k, s, p = case
    d = H.conv_desc(n, c, h, w, m, k, s, p)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((m, c, k, k)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    for bias in (b, None):
        e = H.conv_bias_act(emu_lib, d, x, wt, bias, alpha=0.11, device=None)
        o = H.conv_bias_act(oracle_lib, d, x, wt, bias, alpha=0.11)
        assert H.rel_err(e, o) < TOL


# sae_conv2d_fwd_residual_f32: 1x1 skip convs on every tile (128 / 64 / 32 rows, quad and dword staging, scalar and vector
# epilogue, split-K), odd sizes, stride 2; (n, c, h, w, m, k, stride, pad)
CONV_RESIDUAL = [(2, 40, 8, 8, 70, 1, 1, 0), (1, 70, 16, 16, 130, 1, 1, 0), (2, 33, 7, 9, 20, 1, 1, 0), (1, 256, 4, 4, 40, 1, 1, 0),
                 (3, 12, 12, 20, 24, 1, 1, 0), (2, 33, 7, 7, 130, 1, 2, 0), (1, 20, 15, 15, 36, 1, 2, 0)]


def conv_residual_case(lib, oracle_lib, case, device=None):
    """Against the oracle, and BIT-identical to the same library's conv followed by its add_scale."""
    n, c, h, w, m, k, s, p = case
    d = H.conv_desc(n, c, h, w, m, k, s, p)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((m, c, k, k)).astype(np.float32)
    res = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
    e = H.conv_residual(lib, d, x, wt, res, alpha=0.23, res_scale=0.7071, device=device)
    o = H.conv_residual(oracle_lib, d, x, wt, res, alpha=0.23, res_scale=0.7071)
    assert not np.isnan(e).any() and H.rel_err(e, o) < TOL
    plain = H.conv(lib, 0, d, x, wt, res.shape, alpha=0.23, device=device)
    assert np.array_equal(e, ((plain + res) * np.float32(0.7071)).astype(np.float32))


@pytest.mark.parametrize("case", CONV_RESIDUAL, ids=str)
def test_conv2d_residual(emu_lib, oracle_lib, case):
    conv_residual_case(emu_lib, oracle_lib, case)


GLUE_SHAPES = [(2, 5, 16, 16), (3, 8, 4, 8), (1, 70, 32, 36), (4, 3, 2, 2), (2, 600, 8, 8)]


@pytest.mark.parametrize("shape", GLUE_SHAPES, ids=str)
def test_noise_bias_act(emu_lib, oracle_lib, shape):
    """StyledConv glue: noise + bias + leaky-ReLU forward, its fused backward (bias and noise-weight
    gradients) and the style-modulation backward, against the double-accumulating oracle."""
    rng = np.random.default_rng(21)
    n, c = shape[:2]
    x = rng.standard_normal(shape).astype(np.float32)
    noise = rng.standard_normal((n, 1) + shape[2:]).astype(np.float32)
    nw = np.array([0.37], np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    for nz, bias in ((noise, b), (None, b), (noise, None)):
        e = H.noise_bias_act(emu_lib, x, nz, nw, bias)
        o = H.noise_bias_act(oracle_lib, x, nz, nw, bias)
        assert np.array_equal(e, o)
    y = H.noise_bias_act(oracle_lib, x, noise, nw, b)
    gy = rng.standard_normal(shape).astype(np.float32)
    gx_e, gb_e, gw_e = H.noise_bias_act_bwd(emu_lib, gy, y, noise)
    gx_o, gb_o, gw_o = H.noise_bias_act_bwd(oracle_lib, gy, y, noise)
    assert np.array_equal(gx_e, gx_o)
    assert np.allclose(gb_e, gb_o, rtol=1e-5, atol=1e-4)
    assert np.allclose(gw_e, gw_o, rtol=1e-5, atol=1e-3)
    s = rng.standard_normal((n, c)).astype(np.float32)
    g2_e, gs_e = H.plane_scale_dot(emu_lib, gy, x, s)
    g2_o, gs_o = H.plane_scale_dot(oracle_lib, gy, x, s)
    assert np.array_equal(g2_e, g2_o)
    assert np.allclose(gs_e, gs_o, rtol=1e-5, atol=1e-4)
    # the fused pair (plane_scale_dot, then the producer's noise + bias + activation backward) against the oracle and against
    # the two kernels it replaces, with and without a noise map
    for nz in (noise, None):
        fx, fs, fb, fw = H.plane_scale_dot_act(emu_lib, gy, y, s, nz)
        ox, os_, ob, ow_ = H.plane_scale_dot_act(oracle_lib, gy, y, s, nz)
        assert np.array_equal(fx, ox) and np.allclose(fs, os_, rtol=1e-5, atol=1e-4)
        assert np.allclose(fb, ob, rtol=1e-5, atol=1e-4) and (nz is None or np.allclose(fw, ow_, rtol=1e-5, atol=1e-3))
        t2, s2 = H.plane_scale_dot(emu_lib, gy, y, s)
        x2, b2, w2 = H.noise_bias_act_bwd(emu_lib, t2, y, nz)
        assert np.array_equal(fx, x2) and np.allclose(fs, s2, rtol=1e-6, atol=1e-5) and np.allclose(fb, b2, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("shape", [(5, 3, 3, 3), (7, 300, 1, 1), (3, 64, 3, 3), (2, 2, 1, 1)], ids=str)
def test_weight_demod(emu_lib, oracle_lib, shape):
    """Demodulation factor and the fused weight gradient: the oracle against the reference's own ATen sequence
    (stylegan2_layers.py:290-292) and autograd through it; the kernels against the oracle."""
    import torch
    rng = np.random.default_rng(17)
    w = rng.standard_normal(shape).astype(np.float32)
    geff = rng.standard_normal(shape).astype(np.float32)
    alpha = 1.0 / np.sqrt(np.prod(shape[1:]))
    wt = torch.from_numpy(w).double().requires_grad_()
    weight = alpha * wt                                              # :289
    demod = torch.rsqrt(weight.pow(2).sum([1, 2, 3]) + 1e-8)         # :290-291 (per-sample copies are identical)
    # the conv ran on alpha * d * w; geff = alpha * dL/dW_eff  =>  L = sum(geff / alpha * W_eff)
    loss = (torch.from_numpy(geff).double() / alpha * (weight * demod.view(-1, 1, 1, 1))).sum()
    gw_ref, = torch.autograd.grad(loss, wt)
    d_o = H.weight_demod(oracle_lib, w, alpha)
    assert np.allclose(d_o, demod.detach().numpy(), rtol=3e-7, atol=0)
    gw_o = H.weight_demod_bwd(oracle_lib, geff, w, d_o, alpha)
    assert np.allclose(gw_o, gw_ref.numpy(), rtol=1e-5, atol=1e-6 * float(gw_ref.abs().max()))
    d_e = H.weight_demod(emu_lib, w, alpha)
    assert np.allclose(d_e, d_o, rtol=3e-7, atol=0)
    gw_e = H.weight_demod_bwd(emu_lib, geff, w, d_o, alpha)
    assert np.allclose(gw_e, gw_o, rtol=1e-5, atol=1e-6 * float(np.abs(gw_o).max()))


def test_glue_rejects_odd_planes(emu_lib):
    x = np.zeros((1, 2, 3, 3), np.float32)
    with pytest.raises(Exception):
        H.noise_bias_act(emu_lib, x, None, np.zeros(1, np.float32), None)
    with pytest.raises(Exception):
        H.plane_scale_dot(emu_lib, x, x, np.zeros((1, 2), np.float32))


def _crop_params(rng, k, lo=0.125, hi=0.25):
    flip = np.round(rng.random(k)) * 2 - 1
    scale = rng.random((k, 2)) * (hi - lo) + lo
    offset = (rng.random((k, 2)) * 2 - 1) * (1 - scale)
    return np.stack([flip, scale[:, 0], scale[:, 1], offset[:, 0], offset[:, 1]], 1).astype(np.float32)


@pytest.mark.parametrize("case", [(2, 3, 32, 32, 4, 16, 0.125, 0.25), (1, 2, 20, 28, 3, 9, 0.3, 1.0),
                                  (3, 1, 16, 16, 2, 8, 0.9, 1.0), (2, 3, 48, 40, 5, 24, 0.05, 0.2)], ids=str)
def test_random_crop(emu_lib, oracle_lib, case):
    """Patch sampler: the oracle is pinned to F.grid_sample (what util.apply_random_crop calls,
    util/util.py:338) and to its autograd; the kernels are checked against the oracle."""
    import torch
    import torch.nn.functional as F
    n, c, h, w, crops, size, lo, hi = case
    rng = np.random.default_rng(31)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    params = _crop_params(rng, n * crops, lo, hi)
    gy = rng.standard_normal((n * crops, c, size, size)).astype(np.float32)
    # reference construction of the sampling grid
    lin = torch.linspace(-1.0, 1.0, size)
    pt = torch.from_numpy(params)
    gxs = lin.view(1, 1, size, 1).expand(n * crops, size, size, 1) * pt[:, 0].view(-1, 1, 1, 1)
    gys = lin.view(1, size, 1, 1).expand(n * crops, size, size, 1)
    grid = torch.cat([gxs, gys], 3) * pt[:, 1:3].view(-1, 1, 1, 2) + pt[:, 3:5].view(-1, 1, 1, 2)
    xt = torch.from_numpy(x).requires_grad_()
    xe = xt.unsqueeze(1).expand(-1, crops, -1, -1, -1).flatten(0, 1)
    ref = F.grid_sample(xe, grid, align_corners=False)
    ref.backward(torch.from_numpy(gy))
    o = H.random_crop(oracle_lib, x, params, size, crops)
    assert H.rel_err(o, ref.detach().numpy()) < 2e-6
    ob = H.random_crop_bwd(oracle_lib, gy, params, x.shape, crops)
    assert H.rel_err(ob, xt.grad.numpy()) < 5e-6
    e = H.random_crop(emu_lib, x, params, size, crops)
    assert H.rel_err(e, o) < 2e-6
    eb = H.random_crop_bwd(emu_lib, gy, params, x.shape, crops)
    assert H.rel_err(eb, ob) < 5e-6


@pytest.mark.parametrize("case", [((2, 3, 8, 8), (1, 1, 1, 1)), ((1, 2, 5, 7), (2, 1, 0, 3)), ((2, 1, 2, 2), (1, 1, 1, 1)),
                                  ((1, 4, 16, 16), (1, 2, 1, 2)), ((1, 3, 5, 300), (2, 1, 1, 1)), ((2, 1, 3, 64), (1, 1, 1, 0))], ids=str)
def test_reflect_pad(emu_lib, oracle_lib, case):
    """Reflection pad and its adjoint: oracle pinned to F.pad(mode="reflect") + autograd, kernels to the oracle."""
    import torch
    import torch.nn.functional as F
    shape, pads = case
    rng = np.random.default_rng(41)
    x = rng.standard_normal(shape).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_()
    ref = F.pad(xt, pads, mode="reflect")
    gy = rng.standard_normal(tuple(ref.shape)).astype(np.float32)
    ref.backward(torch.from_numpy(gy))
    o = H.reflect_pad(oracle_lib, x, pads)
    assert np.array_equal(o, ref.detach().numpy())
    oa = H.reflect_pad_adj(oracle_lib, gy, pads)
    assert np.allclose(oa, xt.grad.numpy(), rtol=1e-6, atol=1e-6)
    assert np.array_equal(H.reflect_pad(emu_lib, x, pads), o)
    assert np.allclose(H.reflect_pad_adj(emu_lib, gy, pads), oa, rtol=1e-6, atol=1e-6)
    with pytest.raises(Exception):
        H.reflect_pad(emu_lib, x, (shape[-1], 0, 0, 0))


def test_new_ops_accept_empty_batches_and_reject_bad_geometry(emu_lib, oracle_lib):
    """Edge cases of the glue / sampler / pad entry points: empty inputs are no-ops, bad geometry is an error
    (never a crash), identically in the kernels and the oracle."""
    for lib in (emu_lib, oracle_lib):
        z = np.zeros((0, 3, 4, 4), np.float32)
        assert H.noise_bias_act(lib, z, np.zeros((0, 1, 4, 4), np.float32), np.ones(1, np.float32), np.zeros(3, np.float32)).shape == z.shape
        assert H.reflect_pad(lib, z, (1, 1, 1, 1)).shape == (0, 3, 6, 6)
        assert H.random_crop(lib, z, np.zeros((0, 5), np.float32), 4, 2).shape == (0, 3, 4, 4)
        with pytest.raises(Exception):
            H.reflect_pad(lib, np.zeros((1, 1, 3, 3), np.float32), (3, 0, 0, 0))      # pad >= size
        with pytest.raises(Exception):
            H.random_crop(lib, np.zeros((1, 1, 4, 4), np.float32), np.zeros((1, 5), np.float32), 1, 1)   # size < 2


def test_weight_gradient_slices_follow_the_rounds_of_workgroups(emu_lib):
    """wg_pick_slices (csrc/conv2d.hip): the K slices of a 3x3 weight gradient are priced by rounds of workgroups.  Dpatch's
    256 -> 384 stride-2 layer has 3 x 8 = 24 tiles of 128 x 32: "fill the 256 CUs once" gives 11 slices = 264 workgroups, one
    full round and a second one for eight of them; 10 slices (240) is one round.  The workspace query (a pure host function,
    slices x 9 x Ap x Bp floats) shows the plan."""
    import ctypes as C
    for (n, c, h, m, s, p), slices in [((384, 256, 17, 384, 2, 0), 10), ((384, 384, 8, 384, 1, 1), 7),
                                       ((384, 768, 4, 384, 1, 0), 7), ((16, 128, 64, 128, 1, 1), 64)]:
        d = H.conv_desc(n, c, h, h, m, 3, s, p, False)
        slab = 9 * ((m + 63) // 64 * 64) * ((c + 31) // 32 * 32)
        assert emu_lib.query("conv2d_workspace", C.byref(d), 2) == slices * slab, (n, c, h, m, s)
