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
        rs = (1 + 0.5 * rng.standard_normal(m)).astype(np.float32)
        cs = (1 + 0.5 * rng.standard_normal(c)).astype(np.float32)
        for row, col in [(None, None), (rs, None), (rs, cs)]:
            a = H.wino_weights(lib, w, m, c, sm, sc, flip=flip, alpha=0.37, row_scale=row, col_scale=col, device=dev)
            o = H.wino_weights(oracle_lib, w, m, c, sm, sc, flip=flip, alpha=0.37, row_scale=row, col_scale=col)
            assert H.rel_err(a, o) < 1e-6, (flip, H.rel_err(a, o))
    x = rng.standard_normal((5, 6, 10)).astype(np.float32)
    s = (1 + 0.5 * rng.standard_normal(5)).astype(np.float32)
    for scale, pad in [(None, 1), (s, 1), (s, 0), (None, 2)]:
        a, o = H.wino_input(lib, x, scale, pad=pad, device=dev), H.wino_input(oracle_lib, x, scale, pad=pad)
        assert not np.isnan(a).any() and H.rel_err(a, o) < 1e-6, H.rel_err(a, o)
    md = rng.standard_normal((16, 6, 3, 5)).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    # rows a multiple of 8 wide: four tiles per thread with 16-byte accesses (wino_*4_kernel)
    x8 = rng.standard_normal((3, 8, 16)).astype(np.float32)
    s8 = (1 + 0.5 * rng.standard_normal(3)).astype(np.float32)
    for scale in (None, s8):
        assert H.rel_err(H.wino_input(lib, x8, scale, device=dev), H.wino_input(oracle_lib, x8, scale)) < 1e-6
    md8 = rng.standard_normal((16, 4, 4, 8)).astype(np.float32)
    z8, b8 = rng.standard_normal((2, 8, 16)).astype(np.float32), rng.standard_normal(2).astype(np.float32)
    for kw in [dict(), dict(bias=b8, act=(0.2, 2 ** 0.5)), dict(plane_scale=(1 + 0.5 * rng.standard_normal(4)).astype(np.float32)),
               dict(bias=b8, act=(0.2, 2 ** 0.5), noise=z8, noise_weight=np.array([0.7], np.float32))]:
        assert H.rel_err(H.wino_output(lib, md8, 8, 16, 2, device=dev, **kw), H.wino_output(oracle_lib, md8, 8, 16, 2, **kw)) < 1e-6
    ps = (1 + 0.5 * rng.standard_normal(6)).astype(np.float32)
    z = rng.standard_normal((2, 6, 10)).astype(np.float32)
    zw = np.array([0.7], np.float32)
    for bias, act, scale, noise in [(None, None, None, None), (b, (0.2, 2 ** 0.5), None, None), (None, (0.2, 1.0), ps, None),
                                    (b, (0.2, 2 ** 0.5), None, z), (None, None, ps, None)]:
        kw = dict(bias=bias, act=act, plane_scale=scale, noise=noise, noise_weight=zw if noise is not None else None)
        a = H.wino_output(lib, md, 6, 10, 3, device=dev, **kw)
        o = H.wino_output(oracle_lib, md, 6, 10, 3, **kw)
        assert not np.isnan(a).any() and H.rel_err(a, o) < 1e-6, H.rel_err(a, o)


# n, c, h, w, m, [C, M] weights, pad (1: same; 0: valid -- its data gradient pads by 2; the last one has ONE tile per image, the
# 4 x 4 -> 2 x 2 layer at the end of the patch discriminator)
CASES = [(2, 12, 8, 8, 20, False, 1), (1, 40, 6, 10, 33, False, 1), (3, 9, 4, 4, 70, True, 1), (1, 5, 16, 12, 8, False, 1),
         (1, 6, 16, 16, 10, False, 1),
         (2, 12, 10, 10, 20, False, 0), (5, 24, 4, 4, 16, False, 0), (1, 9, 6, 8, 33, True, 0)]


def _route(lib, oracle_lib, dev):
    rng = np.random.default_rng(37)
    for n, c, h, w, m, cm, pad in CASES:
        d = H.conv_desc(n, c, h, w, m, 3, 1, pad, cm)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = rng.standard_normal((c, m, 3, 3) if cm else (m, c, 3, 3)).astype(np.float32)
        gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
        b = rng.standard_normal(m).astype(np.float32)
        xs = (1 + 0.5 * rng.standard_normal((n, c))).astype(np.float32)
        fwd = H.wino_conv(lib, x, wt, alpha=0.37, cm_layout=cm, pad=pad, device=dev)
        assert H.rel_err(fwd, H.conv(oracle_lib, 0, d, x, wt, gy.shape, alpha=0.37)) < TOL
        dg = H.wino_conv(lib, gy, wt, alpha=0.37, transpose=True, cm_layout=cm, pad=pad, device=dev)
        assert H.rel_err(dg, H.conv(oracle_lib, 1, d, gy, wt, x.shape, alpha=0.37)) < TOL
        if not cm:
            act = H.wino_conv(lib, x, wt, alpha=0.11, bias=b, act=(0.2, 2 ** 0.5), pad=pad, device=dev)
            assert H.rel_err(act, H.conv_bias_act(oracle_lib, d, x, wt, b, alpha=0.11)) < TOL
        # the style-modulated forms against the oracle's modulated convolutions (include/sae_hip.h: sae_conv2d_mod)
        ys = (1 + 0.5 * rng.standard_normal((n, m))).astype(np.float32)
        wm = rng.uniform(0.5, 2, m).astype(np.float32)
        wc = rng.uniform(0.5, 2, c).astype(np.float32)
        mod = H.wino_conv(lib, x, wt, alpha=0.3, x_scale=xs, row_scale=wm, col_scale=wc, cm_layout=cm, pad=pad, device=dev)
        assert H.rel_err(mod, H.modconv(oracle_lib, 0, d, x, wt, gy.shape, x_scale=xs, wm_scale=wm, wc_scale=wc, alpha=0.3)) < TOL
        mdg = H.wino_conv(lib, gy, wt, alpha=0.3, transpose=True, x_scale=ys, row_scale=wc, col_scale=wm, cm_layout=cm, pad=pad, device=dev)
        assert H.rel_err(mdg, H.modconv(oracle_lib, 1, d, gy, wt, x.shape, y_scale=ys, wm_scale=wm, wc_scale=wc, alpha=0.3)) < TOL
        # the weight gradient on the route (plain and with both activation factors) against the oracle's direct one, and its
        # two transforms against the oracle's
        gw, e, gu = H.wino_wgrad(lib, x, gy, alpha=0.37, cm_layout=cm, pad=pad, device=dev)
        assert H.rel_err(gw, H.conv(oracle_lib, 2, d, x, gy, wt.shape, alpha=0.37)) < TOL
        _, eo, guo = H.wino_wgrad(oracle_lib, x, gy, alpha=0.37, cm_layout=cm, pad=pad)
        assert H.rel_err(e, eo) < 1e-6 and H.rel_err(gu, guo) < TOL
        gwm, _, _ = H.wino_wgrad(lib, x, gy, alpha=0.3, cm_layout=cm, x_scale=xs, y_scale=ys, pad=pad, device=dev)
        assert H.rel_err(gwm, H.modconv(oracle_lib, 2, d, x, gy, wt.shape, x_scale=xs, y_scale=ys, alpha=0.3)) < TOL
        if not cm:          # StyledConv's plain form: modulated forward + noise + bias + leaky-ReLU
            z = rng.standard_normal((n, d.oh, d.ow)).astype(np.float32)
            zw = np.array([0.6], np.float32)
            st = H.wino_conv(lib, x, wt, alpha=0.2, x_scale=xs, row_scale=wm, noise=z, noise_weight=zw, bias=b, act=(0.2, 2 ** 0.5),
                             pad=pad, device=dev)
            assert H.rel_err(st, H.modconv_noise_bias_act(oracle_lib, d, x, wt, xs, wm, z, zw, b, alpha=0.2)) < TOL


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


@pytest.mark.parametrize("one_kernel", [False, True])
@pytest.mark.parametrize("pad", [1, 0])
def test_python_route_through_autograd(oracle_lib, monkeypatch, pad, one_kernel):
    """stylegan2_op.winograd behind conv2d_gemm: forward with the fused activation, data gradient and weight gradient take the
    route (three-kernel form / the ONE-kernel form for forward and data gradient) and agree with the direct kernels."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as G, winograd
    monkeypatch.setattr(hip_lib, "_LIB", oracle_lib)
    torch.manual_seed(3)
    side = 8 if pad else 10
    x = torch.randn(2, 12, side, side, requires_grad=True)
    w = torch.randn(16, 12, 3, 3, requires_grad=True)
    b = torch.randn(16, requires_grad=True)
    geom = G._Geom(2, 12, side, side, 16, 3, 1, pad, False, 0.25)

    def run():
        y = G.ConvBiasAct.apply(x, w, b, geom, 0.2, 2 ** 0.5)
        gx, gw, gb = torch.autograd.grad((y * y).sum(), (x, w, b))
        return y.detach(), gx, gw, gb

    with winograd.override(enabled=False):
        assert not winograd.eligible(geom)
        direct = run()
    calls = []
    orig = winograd.conv
    monkeypatch.setattr(winograd, "conv", lambda *a, **k: (calls.append(k.get("transpose", False)), orig(*a, **k))[1])
    wg_calls = []
    orig_wg = winograd.wgrad
    monkeypatch.setattr(winograd, "wgrad", lambda *a, **k: (wg_calls.append(1), orig_wg(*a, **k))[1])
    with winograd.override(enabled=True, min_c=8, fused=one_kernel):
        assert winograd.route(geom, winograd.FWD) == ("fused" if one_kernel else "unfused")
        assert winograd.route(geom, winograd.WGRAD) == "unfused"
        routed = run()
    assert calls == [False, True], calls          # the fused forward, then the data gradient
    assert wg_calls == [1], wg_calls              # ... and the weight gradient
    for a, o in zip(routed, direct):
        assert float((a - o).abs().max() / o.abs().max()) < TOL


@pytest.mark.parametrize("one_kernel", [False, True])
def test_modulated_nodes_take_the_route(oracle_lib, monkeypatch, one_kernel):
    """ModulatedConv (forward: x_scale + demodulation; backward: the data gradient with the factors on the transposed axes) and the
    fused StyledConv forward (noise + bias + activation in the output transform) behind SAE_WINOGRAD=1 against the direct path."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as G, winograd
    monkeypatch.setattr(hip_lib, "_LIB", oracle_lib)
    torch.manual_seed(5)
    x = torch.randn(2, 12, 8, 8, requires_grad=True)
    s = (1 + 0.3 * torch.randn(2, 12)).requires_grad_(True)
    w = torch.randn(16, 12, 3, 3, requires_grad=True)
    noise = torch.randn(2, 1, 8, 8)
    nw = torch.tensor([0.4], requires_grad=True)
    b = torch.randn(16, requires_grad=True)

    def run():
        y1 = G.modulated_conv2d(x, s, w, padding=1, alpha=0.1, demod_eps=1e-8)
        y2 = G.styled_modulated_conv2d(x, s, w, noise, nw, b, padding=1, alpha=0.1, demod_eps=1e-8)
        grads = torch.autograd.grad((y1 * y1).sum() + (y2 * y2).sum(), (x, s, w, nw, b))
        return (y1.detach(), y2.detach()) + grads

    with winograd.override(enabled=False):
        direct = run()
    calls = []
    orig = winograd.conv
    monkeypatch.setattr(winograd, "conv", lambda *a, **k: (calls.append(k.get("transpose", False)), orig(*a, **k))[1])
    with winograd.override(enabled=True, min_c=8, fused=one_kernel):
        routed = run()
    assert calls.count(False) == 2 and calls.count(True) == 2, calls      # two forwards, two data gradients
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
    assert H.rel_err(H.wino_fused_conv(lib, x, wt, device="cuda:0"), direct) < TOL


@pytest.mark.gpu
def test_fused_route_on_the_gpu(oracle_lib):
    """The ONE-kernel route against the oracle's direct convolution (all epilogues, modulated forms, both paddings, the data
    gradient), then two layers of the step's size against the direct MFMA kernels: 128 -> 128 @64^2 (a partial last channel block
    is not involved, 16 chunks) and 200 -> 72 @32^2 (channel counts that are no multiple of the chunk or the block)."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    lib = hip_lib.get()
    _fused_route(lib, oracle_lib, "cuda:0")
    rng = np.random.default_rng(43)
    for n, c, m, side in [(3, 128, 128, 64), (2, 200, 72, 32)]:
        x = rng.standard_normal((n, c, side, side)).astype(np.float32)
        wt = (rng.standard_normal((m, c, 3, 3)) / (3 * c ** 0.5)).astype(np.float32)
        gy = rng.standard_normal((n, m, side, side)).astype(np.float32)
        d = H.conv_desc(n, c, side, side, m, 3, 1, 1)
        direct = H.conv(lib, 0, d, x, wt, gy.shape, alpha=1.0, device="cuda:0")
        assert H.rel_err(H.wino_fused_conv(lib, x, wt, device="cuda:0"), direct) < TOL
        dgrad = H.conv(lib, 1, d, gy, wt, x.shape, alpha=1.0, device="cuda:0")
        assert H.rel_err(H.wino_fused_conv(lib, gy, wt, transpose=True, device="cuda:0"), dgrad) < TOL


def test_train_step_with_the_route(oracle_lib, monkeypatch):
    """Four optimiser calls of the micro preset (D, G, D + lazy R1 -- the second-order path --, G) with every eligible layer on the
    route against the same calls on the direct kernels: the losses agree to the rounding of the route."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import parity_common as P
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import winograd
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import SwappingAutoencoderOptimizer
    monkeypatch.setattr(hip_lib, "_LIB", oracle_lib)

    def run(route):
        prev = winograd.configure(enabled=route, min_c=4)
        calls = []
        orig = winograd.conv
        monkeypatch.setattr(winograd, "conv", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        torch.manual_seed(0)
        opt, model, net = P.build_micro("cpu", batch_size=4)
        opt.R1_once_every = 2
        optimizer = SwappingAutoencoderOptimizer(model, fused_adam=True)
        out = []
        for i in range(4):
            torch.manual_seed(100 + i)
            x = torch.rand(4, 3, 32, 32) * 2 - 1
            out.append({k: float(v) for k, v in optimizer.train_one_step({"real_A": x}, i).items()})
        monkeypatch.setattr(winograd, "conv", orig)
        winograd.configure(*prev)
        return out, len(calls)

    direct, n0 = run(False)
    routed, n1 = run(True)
    assert n0 == 0 and n1 > 20, (n0, n1)
    assert any("D_R1" in c for c in routed)
    for a, b in zip(routed, direct):
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-4 * max(1.0, abs(b[k])), (k, a[k], b[k])


def test_wide_rows_take_the_four_tile_kernels():
    """Rows a multiple of 8 wide go through wino_input4 / wino_gy4 / wino_output4 (16-byte accesses); the dispatch trace of the
    emulator build shows it (a subprocess: the trace knob is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, abi_harness as H\n"
            "from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary\n"
            "from emu import build_emu\n"
            "lib = SaeLibrary(build_emu.build(), prefix='sae_', device_only=False)\n"
            "x = np.ones((2, 8, 16, 16), np.float32); w = np.ones((8, 8, 3, 3), np.float32)\n"
            "H.wino_conv(lib, x, w); H.wino_wgrad(lib, x, x)\n"
            "H.wino_conv(lib, x[:, :, :6, :10].copy(), w)\n" % (root, os.path.join(root, "tests")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SAE_TRACE_DISPATCH="1"), capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stderr.splitlines() if l.startswith("sae-dispatch wino")]
    assert lines == ["sae-dispatch wino input4", "sae-dispatch wino output4", "sae-dispatch wino input4", "sae-dispatch wino gy4"], lines


# ---- the ONE-kernel route (csrc/winograd_fused.hip): n, c, h, w, m, [C, M] weights, pad.  Channel counts around the 8-channel
# chunk and the 64-channel block, maps that fill a 4 x 16 tile block, spill over it, or leave most of it empty, several images per block
FUSED_CASES = [(2, 12, 8, 8, 20, False, 1), (1, 40, 6, 10, 70, False, 1), (3, 9, 4, 4, 70, True, 1), (1, 5, 16, 36, 8, False, 1),
               (2, 16, 10, 10, 20, False, 0), (5, 24, 4, 4, 16, False, 0), (1, 9, 6, 8, 33, True, 0), (1, 8, 40, 12, 64, False, 1),
               (1, 8, 8, 100, 16, False, 1),      # 50 tiles per row: four blocks, the middle two touch no border (the select-free loop)
               (3, 10, 20, 40, 12, False, 1)]     # 3 images x 3 block rows x 2 block columns (16 x 4 tiles each, ragged in both directions)


def _fused_route(lib, oracle_lib, dev):
    rng = np.random.default_rng(41)
    for n, c, h, w, m, cm, pad in FUSED_CASES:
        d = H.conv_desc(n, c, h, w, m, 3, 1, pad, cm)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = rng.standard_normal((c, m, 3, 3) if cm else (m, c, 3, 3)).astype(np.float32)
        gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
        b = rng.standard_normal(m).astype(np.float32)
        xs = (1 + 0.5 * rng.standard_normal((n, c))).astype(np.float32)
        ys = (1 + 0.5 * rng.standard_normal((n, m))).astype(np.float32)
        wm = rng.uniform(0.5, 2, m).astype(np.float32)
        wc = rng.uniform(0.5, 2, c).astype(np.float32)
        case = (n, c, h, w, m, cm, pad)
        fwd = H.wino_fused_conv(lib, x, wt, alpha=0.37, cm_layout=cm, pad=pad, device=dev)
        assert not np.isnan(fwd).any() and H.rel_err(fwd, H.conv(oracle_lib, 0, d, x, wt, gy.shape, alpha=0.37)) < TOL, case
        if d.ow < 4 and lib is not oracle_lib:      # the kernel fetches patch rows as 16-byte loads: rows of at least 4 floats, else refused
            with pytest.raises(Exception):
                H.wino_fused_conv(lib, gy, wt, alpha=0.37, transpose=True, cm_layout=cm, pad=pad, device=dev)
            continue
        dg = H.wino_fused_conv(lib, gy, wt, alpha=0.37, transpose=True, cm_layout=cm, pad=pad, device=dev)
        assert not np.isnan(dg).any() and H.rel_err(dg, H.conv(oracle_lib, 1, d, gy, wt, x.shape, alpha=0.37)) < TOL, case
        mod = H.wino_fused_conv(lib, x, wt, alpha=0.3, x_scale=xs, row_scale=wm, col_scale=wc, cm_layout=cm, pad=pad, device=dev)
        assert H.rel_err(mod, H.modconv(oracle_lib, 0, d, x, wt, gy.shape, x_scale=xs, wm_scale=wm, wc_scale=wc, alpha=0.3)) < TOL, case
        mdg = H.wino_fused_conv(lib, gy, wt, alpha=0.3, transpose=True, x_scale=ys, row_scale=wc, col_scale=wm, cm_layout=cm, pad=pad,
                                device=dev)
        assert H.rel_err(mdg, H.modconv(oracle_lib, 1, d, gy, wt, x.shape, y_scale=ys, wm_scale=wm, wc_scale=wc, alpha=0.3)) < TOL, case
        if not cm:
            act = H.wino_fused_conv(lib, x, wt, alpha=0.11, bias=b, act=(0.2, 2 ** 0.5), pad=pad, device=dev)
            assert H.rel_err(act, H.conv_bias_act(oracle_lib, d, x, wt, b, alpha=0.11)) < TOL, case
            z = rng.standard_normal((n, d.oh, d.ow)).astype(np.float32)
            zw = np.array([0.6], np.float32)
            st = H.wino_fused_conv(lib, x, wt, alpha=0.2, x_scale=xs, row_scale=wm, noise=z, noise_weight=zw, bias=b,
                                   act=(0.2, 2 ** 0.5), pad=pad, device=dev)
            assert H.rel_err(st, H.modconv_noise_bias_act(oracle_lib, d, x, wt, xs, wm, z, zw, b, alpha=0.2)) < TOL, case
        # ... and the entry point against its own restatement in the oracle (same contract, incl. out_scale)
        os_ = (1 + 0.5 * rng.standard_normal((n, m))).astype(np.float32)
        a = H.wino_fused_conv(lib, x, wt, alpha=0.5, out_scale=os_, cm_layout=cm, pad=pad, device=dev)
        o = H.wino_fused_conv(oracle_lib, x, wt, alpha=0.5, out_scale=os_, cm_layout=cm, pad=pad)
        assert H.rel_err(a, o) < TOL, case


def test_fused_route_equals_the_direct_convolution_on_the_emulator(emu_lib, oracle_lib):
    _fused_route(emu_lib, oracle_lib, None)


def test_oracle_fused_route_equals_the_oracle_direct_convolution(oracle_lib):
    _fused_route(oracle_lib, oracle_lib, None)


# n, c, h, w, m, [C, M] weights, pad: output rows a multiple of 16 pixels (chunks of 8 tiles); channel counts around the 64 block;
# pixel counts that give one slice, several slices, and a ragged last slice
FUSED_WGRAD_CASES = [(1, 12, 16, 16, 20, False, 1), (3, 70, 8, 16, 33, False, 1), (2, 9, 18, 18, 70, True, 0),
                     (5, 64, 32, 32, 64, False, 1), (2, 5, 4, 34, 130, False, 0)]


def _fused_wgrad(lib, oracle_lib, dev):
    rng = np.random.default_rng(47)
    for n, c, h, w, m, cm, pad in FUSED_WGRAD_CASES:
        d = H.conv_desc(n, c, h, w, m, 3, 1, pad, cm)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
        xs = (1 + 0.5 * rng.standard_normal((n, c))).astype(np.float32)
        ys = (1 + 0.5 * rng.standard_normal((n, m))).astype(np.float32)
        shape = (c, m, 3, 3) if cm else (m, c, 3, 3)
        case = (n, c, h, w, m, cm, pad)
        gw = H.wino_fused_wgrad(lib, x, gy, alpha=0.37, cm_layout=cm, pad=pad, device=dev)
        assert not np.isnan(gw).any() and H.rel_err(gw, H.conv(oracle_lib, 2, d, x, gy, shape, alpha=0.37)) < TOL, case
        gwm = H.wino_fused_wgrad(lib, x, gy, alpha=0.3, cm_layout=cm, x_scale=xs, y_scale=ys, pad=pad, device=dev)
        assert H.rel_err(gwm, H.modconv(oracle_lib, 2, d, x, gy, shape, x_scale=xs, y_scale=ys, alpha=0.3)) < TOL, case
        gwx = H.wino_fused_wgrad(lib, x, gy, alpha=0.3, cm_layout=cm, x_scale=xs, pad=pad, device=dev)
        assert H.rel_err(gwx, H.modconv(oracle_lib, 2, d, x, gy, shape, x_scale=xs, alpha=0.3)) < TOL, case
    with pytest.raises(Exception):          # rows of 12 pixels: no whole chunks of 8 tiles
        H.wino_fused_wgrad(lib, np.zeros((1, 4, 12, 12), np.float32), np.zeros((1, 4, 12, 12), np.float32), device=dev)


def test_fused_weight_gradient_on_the_emulator(emu_lib, oracle_lib):
    _fused_wgrad(emu_lib, oracle_lib, None)


def test_oracle_fused_weight_gradient_equals_the_oracle_direct_one(oracle_lib):
    _fused_wgrad(oracle_lib, oracle_lib, None)


@pytest.mark.gpu
def test_fused_weight_gradient_on_the_gpu(oracle_lib):
    from swapping_autoencoder_pytorch_amd import hip_lib
    lib = hip_lib.get()
    _fused_wgrad(lib, oracle_lib, "cuda:0")
    rng = np.random.default_rng(53)
    for n, c, m, side in [(3, 128, 128, 64), (2, 200, 72, 32)]:          # against the direct MFMA weight gradient
        x = rng.standard_normal((n, c, side, side)).astype(np.float32)
        gy = rng.standard_normal((n, m, side, side)).astype(np.float32)
        d = H.conv_desc(n, c, side, side, m, 3, 1, 1)
        direct = H.conv(lib, 2, d, x, gy, (m, c, 3, 3), alpha=0.01, device="cuda:0")
        assert H.rel_err(H.wino_fused_wgrad(lib, x, gy, alpha=0.01, device="cuda:0"), direct) < TOL


def test_python_route_one_kernel_weight_gradient(oracle_lib, monkeypatch):
    """A map 16 pixels wide: forward, data gradient AND weight gradient on the one-kernel forms (winograd.route), plain and
    style-modulated, against the direct kernels -- incl. writing into a given gradient buffer."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as G, winograd
    monkeypatch.setattr(hip_lib, "_LIB", oracle_lib)
    torch.manual_seed(9)
    x = torch.randn(2, 12, 16, 16, requires_grad=True)
    w = torch.randn(16, 12, 3, 3, requires_grad=True)
    b = torch.randn(16, requires_grad=True)
    s = (1 + 0.3 * torch.randn(2, 12)).requires_grad_(True)
    geom = G._Geom(2, 12, 16, 16, 16, 3, 1, 1, False, 0.25)

    def run():
        y = G.ConvBiasAct.apply(x, w, b, geom, 0.2, 2 ** 0.5)
        y2 = G.modulated_conv2d(x, s, w, padding=1, alpha=0.1, demod_eps=1e-8)
        return (y.detach(), y2.detach()) + torch.autograd.grad((y * y).sum() + (y2 * y2).sum(), (x, w, b, s))

    with winograd.override(enabled=False):
        direct = run()
    with winograd.override(enabled=True, min_c=8, fused=True):
        assert [winograd.route(geom, op) for op in (winograd.FWD, winograd.DGRAD, winograd.WGRAD)] == ["fused"] * 3
        routed = run()
        out = torch.full((16, 12, 3, 3), float("nan"))
        gy = torch.randn(2, 16, 16, 16)
        got = winograd.wgrad(x.detach(), gy, geom, out=out)
        assert got.data_ptr() == out.data_ptr() and not torch.isnan(out).any()
    for a, o in zip(routed, direct):
        assert float((a - o).abs().max() / o.abs().max()) < TOL


def test_route_keeps_activations_over_2_gib_off_the_one_kernel_forms(oracle_lib, monkeypatch):
    """The one-kernel kernels address x / gy with 32-bit byte offsets (the C-ABI refuses larger tensors): the router must not pick
    them there, whatever the estimates say."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as G, winograd
    monkeypatch.setattr(hip_lib, "_LIB", oracle_lib)
    big = G._Geom(64, 256, 512, 512, 256, 3, 1, 1, False, 1.0)          # 17 GB of input
    ok = G._Geom(16, 256, 128, 128, 256, 3, 1, 1, False, 1.0)
    with winograd.override(enabled=True):
        assert [winograd.route(ok, op) for op in (winograd.FWD, winograd.DGRAD, winograd.WGRAD)] == ["fused"] * 3
        assert all(winograd.route(big, op) != "fused" for op in (winograd.FWD, winograd.DGRAD, winograd.WGRAD))


def test_fused_conv_never_reads_past_the_input_on_the_emulator(emu_lib, oracle_lib):
    """The channel plane of wino_fused_kernel's input loads rides in the buffer load's SCALAR offset; a right-border window in the
    last row of the last image reaches 4 (pad 1) or 8 (pad 2, the data gradient) bytes past x for every channel but the first
    unless the range check stops it.  It does: gfx950 checks voffset against num_records - soffset (measured in round 6, when a
    kernel that subtracted the scalar offset from num_records itself read zeros for the last image), and the emulator models
    that rule.  The emulator counts loads that pass their range check and still touch the 64 bytes after x."""
    import ctypes
    rng = np.random.default_rng(5)
    dll = emu_lib._dll
    dll.hipemu_set_guard.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    dll.hipemu_guard_hits.restype = ctypes.c_longlong
    for n, c, h, w, m, ipad in [(2, 12, 8, 8, 16, 1), (1, 9, 6, 10, 70, 2), (3, 8, 16, 32, 8, 1)]:
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = (0.3 * rng.standard_normal((m, c, 3, 3))).astype(np.float32)
        # x at the front of a larger buffer: the floats behind it are NaN, so a read that escaped the guard would also show
        store = np.full(x.size + 16, np.nan, np.float32)
        store[:x.size] = x.reshape(-1)
        floats = emu_lib.query("wino_fused_weights_floats", m, c)
        u = np.zeros(floats, np.float32)
        emu_lib.call("wino_fused_weights_f32", wt.ctypes.data, None, None, u.ctypes.data, m, c, c * 9, 9, 0, 1.0, None)
        oh, ow = h + 2 * ipad - 2, w + 2 * ipad - 2
        y = np.full((n, m, oh, ow), np.nan, np.float32)
        end = store.ctypes.data + 4 * x.size
        dll.hipemu_set_guard(end, end + 64)
        try:
            emu_lib.call("wino_fused_conv_f32", store.ctypes.data, None, u.ctypes.data, None, None, None, None, y.ctypes.data,
                         n, c, m, h, w, ipad, 0, 0.0, 1.0, None)
            hits = dll.hipemu_guard_hits()
        finally:
            dll.hipemu_set_guard(None, None)
        assert hits == 0, (n, c, h, w, m, ipad, hits)
        d = H.conv_desc(n, c, h, w, m, 3, 1, ipad, False)
        assert H.rel_err(y, H.conv(oracle_lib, 0, d, x, wt, y.shape)) < TOL
