"""The stride-2 3x3 family on its polyphase minimal-filtering form (csrc/s2wino.hip, include/sae_hip.h: sae_s2wino_*) against the
oracle's DIRECT convolution (oracle/sae_oracle.c: oracle_conv2d_*): on the emulator build of the unmodified kernel source here,
through the C-ABI of libsae_hip.so on the GPU.  Tolerance: 2e-5 of the result's largest magnitude (fp32 accumulation over
<= 9 * 512 products; the north star asks for 1e-4)."""
import numpy as np
import pytest

from . import abi_harness as H

TOL = 2e-5

# n, m (gradient channels), c (data channels), h, w of the SMALL side, [C, M] weight layout.  Channel counts around the 8-channel
# chunk and the 64-channel block; maps of one tile, of a whole 4 x 16 block, ragged blocks, several images per block, strips longer
# than one block
DGRAD_CASES = [(2, 12, 20, 4, 4, False), (1, 40, 70, 6, 10, False), (3, 9, 70, 2, 4, True), (1, 5, 8, 16, 36, False),
               (2, 16, 20, 8, 8, True), (1, 8, 64, 40, 12, False), (1, 8, 16, 8, 100, False), (3, 10, 12, 20, 40, False),
               (70, 3, 5, 2, 4, False)]


def _dgrad(lib, oracle_lib, dev, cases=DGRAD_CASES):
    rng = np.random.default_rng(61)
    for n, m, c, h, w, cm in cases:
        d = H.conv_desc(n, c, 2 * h + 1, 2 * w + 1, m, 3, 2, 0, cm)
        assert (d.oh, d.ow) == (h, w)
        gy = rng.standard_normal((n, m, h, w)).astype(np.float32)
        wt = rng.standard_normal((c, m, 3, 3) if cm else (m, c, 3, 3)).astype(np.float32)
        ys = (1 + 0.5 * rng.standard_normal((n, m))).astype(np.float32)
        wm = rng.uniform(0.5, 2, m).astype(np.float32)
        wc = rng.uniform(0.5, 2, c).astype(np.float32)
        case = (n, m, c, h, w, cm)
        xs = (n, c, 2 * h + 1, 2 * w + 1)
        dg = H.s2wino_dgrad(lib, gy, wt, alpha=0.37, cm_layout=cm, device=dev)
        ref = H.conv(oracle_lib, 1, d, gy, wt, xs, alpha=0.37)
        assert not np.isnan(dg).any() and H.rel_err(dg, ref) < TOL, (case, H.rel_err(dg, ref))
        mdg = H.s2wino_dgrad(lib, gy, wt, alpha=0.3, cm_layout=cm, g_scale=ys, row_scale=wc, col_scale=wm, device=dev)
        assert H.rel_err(mdg, H.modconv(oracle_lib, 1, d, gy, wt, xs, y_scale=ys, wm_scale=wm, wc_scale=wc, alpha=0.3)) < TOL, case
        os_ = (1 + 0.5 * rng.standard_normal((n, c))).astype(np.float32)
        a = H.s2wino_dgrad(lib, gy, wt, alpha=0.5, cm_layout=cm, out_scale=os_, device=dev)
        o = H.s2wino_dgrad(oracle_lib, gy, wt, alpha=0.5, cm_layout=cm, out_scale=os_)
        assert H.rel_err(a, o) < TOL, case
    if lib is not oracle_lib:
        with pytest.raises(Exception):          # odd small side: not this kernel's problem
            H.s2wino_dgrad(lib, np.zeros((1, 4, 3, 4), np.float32), np.zeros((4, 4, 3, 3), np.float32), device=dev)


def test_oracle_polyphase_data_gradient_equals_the_oracle_direct_one(oracle_lib):
    _dgrad(oracle_lib, oracle_lib, None)


def test_polyphase_data_gradient_on_the_emulator(emu_lib, oracle_lib):
    _dgrad(emu_lib, oracle_lib, None)


@pytest.mark.gpu
def test_polyphase_data_gradient_on_the_gpu(oracle_lib):
    from swapping_autoencoder_pytorch_amd import hip_lib
    lib = hip_lib.get()
    _dgrad(lib, oracle_lib, "cuda:0")
    rng = np.random.default_rng(67)
    for n, m, c, side in [(3, 128, 64, 32), (2, 72, 200, 16)]:          # against the direct MFMA data gradient
        gy = rng.standard_normal((n, m, side, side)).astype(np.float32)
        wt = rng.standard_normal((m, c, 3, 3)).astype(np.float32)
        d = H.conv_desc(n, c, 2 * side + 1, 2 * side + 1, m, 3, 2, 0)
        direct = H.conv(lib, 1, d, gy, wt, (n, c, 2 * side + 1, 2 * side + 1), alpha=0.01, device="cuda:0")
        assert H.rel_err(H.s2wino_dgrad(lib, gy, wt, alpha=0.01, device="cuda:0"), direct) < TOL


def test_python_route_polyphase_data_gradient(oracle_lib, monkeypatch):
    """The route of stylegan2_op/winograd.py: the data gradient of a stride-2 ConvLayer's conv and the forward of the stride-2
    transposed (modulated, demodulated) conv on the polyphase form, against the direct kernels -- forward values and every
    gradient the autograd nodes produce (the weight gradient and the forward stay direct)."""
    import torch
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as G, winograd
    monkeypatch.setattr(hip_lib, "_LIB", oracle_lib)
    torch.manual_seed(11)
    x = torch.randn(2, 12, 17, 17, requires_grad=True)          # ConvLayer(downsample): blur output 17 x 17 -> 8 x 8
    w = torch.randn(16, 12, 3, 3, requires_grad=True)
    b = torch.randn(16, requires_grad=True)
    z = torch.randn(2, 16, 8, 8, requires_grad=True)            # ModulatedConv2d(upsample): 8 x 8 -> 17 x 17
    wu = torch.randn(12, 16, 3, 3, requires_grad=True)
    s = (1 + 0.3 * torch.randn(2, 16)).requires_grad_(True)
    geom = G._Geom(2, 12, 17, 17, 16, 3, 2, 0, False, 0.25)

    def run():
        y = G.conv2d_bias_act(x, w, b, stride=2, padding=0, alpha=0.25)
        up = G.modulated_conv2d(z, s, wu, alpha=0.1, transposed=True, demod_eps=1e-8)
        return (y.detach(), up.detach()) + torch.autograd.grad((y * y).sum() + (up * up).sum(), (x, w, b, z, wu, s))

    with winograd.override(enabled=False):
        direct = run()
    with winograd.override(enabled=True, min_c=8):
        assert [winograd.route(geom, op) for op in (winograd.FWD, winograd.DGRAD, winograd.WGRAD)] == [None, "s2poly", None]
        routed = run()
    for a, o in zip(routed, direct):
        assert float((a - o).abs().max() / o.abs().max()) < TOL
    with winograd.override(enabled=True):      # the measured rule: small-side maps of 8 .. 32 with >= 256 contraction channels
        assert winograd.route(G._Geom(40, 512, 65, 65, 512, 3, 2, 0, False, 1.0), winograd.DGRAD) == "s2poly"
        assert winograd.route(G._Geom(40, 128, 257, 257, 256, 3, 2, 0, False, 1.0), winograd.DGRAD) is None
        assert winograd.route(G._Geom(40, 512, 64, 64, 512, 3, 2, 0, False, 1.0), winograd.DGRAD) is None      # not a 2^k + 1 map
