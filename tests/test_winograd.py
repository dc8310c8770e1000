"""Winograd F(2x2, 3x3) route (csrc/winograd.hip, stylegan2_op/winograd.py): the three transforms against the oracle's
restatement from the published matrices, and the whole route -- transforms + sixteen 1x1 convolutions -- against the DIRECT 3x3
convolution of the oracle (forward, data gradient, fused bias + leaky-ReLU, a style-modulated input, [C, M] weights).
On the emulator here, on the GPU with -m gpu."""
import numpy as np
import pytest
import torch

import abi_harness as H

TOL = 2e-5      # the kernel tests' bound; the route's own rounding is ~1e-6


def _transforms(lib, oracle_lib, dev):
    rng = np.random.default_rng(31)
    w = rng.standard_normal((10, 7, 3, 3)).astype(np.float32)
    for flip, (m, c, sm, sc) in [(False, (10, 7, 63, 9)), (True, (7, 10, 9, 63))]:
        a = H.wino_weights(lib, w, m, c, sm, sc, flip=flip, alpha=0.37, device=dev)
        o = H.wino_weights(oracle_lib, w, m, c, sm, sc, flip=flip, alpha=0.37)
        assert H.rel_err(a, o) < 1e-6, (flip, H.rel_err(a, o))
    x = rng.standard_normal((5, 6, 10)).astype(np.float32)
    s = (1 + 0.5 * rng.standard_normal(5)).astype(np.float32)
    for scale in (None, s):
        a, o = H.wino_input(lib, x, scale, device=dev), H.wino_input(oracle_lib, x, scale)
        assert not np.isnan(a).any() and H.rel_err(a, o) < 1e-6, H.rel_err(a, o)
    md = rng.standard_normal((16, 6, 3, 5)).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    for bias, act in [(None, None), (b, (0.2, 2 ** 0.5)), (None, (0.2, 1.0))]:
        a = H.wino_output(lib, md, 6, 10, 3, bias=bias, act=act, device=dev)
        o = H.wino_output(oracle_lib, md, 6, 10, 3, bias=bias, act=act)
        assert not np.isnan(a).any() and H.rel_err(a, o) < 1e-6, H.rel_err(a, o)


CASES = [(2, 12, 8, 8, 20, False), (1, 40, 6, 10, 33, False), (3, 9, 4, 4, 70, True), (1, 5, 16, 12, 8, False)]


def _route(lib, oracle_lib, dev):
    rng = np.random.default_rng(37)
    for n, c, h, w, m, cm in CASES:
        d = H.conv_desc(n, c, h, w, m, 3, 1, 1, cm)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = rng.standard_normal((c, m, 3, 3) if cm else (m, c, 3, 3)).astype(np.float32)
        gy = rng.standard_normal((n, m, h, w)).astype(np.float32)
        b = rng.standard_normal(m).astype(np.float32)
        xs = (1 + 0.5 * rng.standard_normal((n, c))).astype(np.float32)
        fwd = H.wino_conv(lib, x, wt, alpha=0.37, cm_layout=cm, device=dev)
        assert H.rel_err(fwd, H.conv(oracle_lib, 0, d, x, wt, gy.shape, alpha=0.37)) < TOL
        dg = H.wino_conv(lib, gy, wt, alpha=0.37, transpose=True, cm_layout=cm, device=dev)
        assert H.rel_err(dg, H.conv(oracle_lib, 1, d, gy, wt, x.shape, alpha=0.37)) < TOL
        if not cm:
            act = H.wino_conv(lib, x, wt, alpha=0.11, bias=b, act=(0.2, 2 ** 0.5), device=dev)
            assert H.rel_err(act, H.conv_bias_act(oracle_lib, d, x, wt, b, alpha=0.11)) < TOL
        mod = H.wino_conv(lib, x, wt, alpha=0.3, x_scale=xs, cm_layout=cm, device=dev)
        ref = H.conv(oracle_lib, 0, d, x * xs[:, :, None, None], wt, gy.shape, alpha=0.3)
        assert H.rel_err(mod, ref) < TOL


def test_transforms_on_the_emulator(emu_lib, oracle_lib):
    _transforms(emu_lib, oracle_lib, None)


def test_route_equals_the_direct_convolution_on_the_emulator(emu_lib, oracle_lib):
    _route(emu_lib, oracle_lib, None)


def test_oracle_route_equals_the_oracle_direct_convolution(oracle_lib):
    """The restatement from the published matrices is a convolution: pins the oracle's own transforms."""
    _route(oracle_lib, oracle_lib, None)


def test_bad_geometry_is_refused(emu_lib):
    x = np.zeros((2, 5, 6), np.float32)
    with pytest.raises(Exception):
        H.wino_input(emu_lib, x)           # odd height: no whole 2x2 output tiles


def test_python_route_through_autograd(oracle_lib, monkeypatch):
    """stylegan2_op.winograd behind conv2d_gemm (SAE_WINOGRAD=1): forward, data gradient and the fused activation take the route
    and agree with the direct kernels; the weight gradient stays the direct kernel."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as G, winograd
    monkeypatch.setattr(hip_lib, "_LIB", oracle_lib)
    torch.manual_seed(3)
    x = torch.randn(2, 12, 8, 8, requires_grad=True)
    w = torch.randn(16, 12, 3, 3, requires_grad=True)
    b = torch.randn(16, requires_grad=True)
    geom = G._Geom(2, 12, 8, 8, 16, 3, 1, 1, False, 0.25)

    def run():
        y = G.ConvBiasAct.apply(x, w, b, geom, 0.2, 2 ** 0.5)
        gx, gw, gb = torch.autograd.grad((y * y).sum(), (x, w, b))
        return y.detach(), gx, gw, gb

    monkeypatch.setenv("SAE_WINOGRAD", "0")
    direct = run()
    monkeypatch.setenv("SAE_WINOGRAD", "1")
    monkeypatch.setenv("SAE_WINOGRAD_MIN_C", "8")
    assert winograd.eligible(geom)
    calls = []
    orig = winograd.conv
    monkeypatch.setattr(winograd, "conv", lambda *a, **k: (calls.append(k.get("transpose", False)), orig(*a, **k))[1])
    routed = run()
    assert calls == [False, True], calls          # the fused forward, then the data gradient
    for a, o in zip(routed, direct):
        assert float((a - o).abs().max() / o.abs().max()) < TOL


@pytest.mark.gpu
def test_transforms_and_route_on_the_gpu(oracle_lib):
    from swapping_autoencoder_pytorch_amd import hip_lib
    lib = hip_lib.get()
    _transforms(lib, oracle_lib, "cuda:0")
    _route(lib, oracle_lib, "cuda:0")
    # a layer of the step's size: 512 -> 512 @32^2, 4 images
    rng = np.random.default_rng(41)
    x = rng.standard_normal((4, 512, 32, 32)).astype(np.float32)
    wt = (rng.standard_normal((512, 512, 3, 3)) / 68).astype(np.float32)
    d = H.conv_desc(4, 512, 32, 32, 512, 3, 1, 1)
    direct = H.conv(lib, 0, d, x, wt, (4, 512, 32, 32), alpha=1.0, device="cuda:0")
    assert H.rel_err(H.wino_conv(lib, x, wt, device="cuda:0"), direct) < TOL
