"""GPU, BASELINE sizes (church preset, 256x256, B = 16): size-independent properties of the hot-path
kernels, where the CPU oracle would take hours.  Adjointness <A x, g> = <x, A^T g> ties forward,
dgrad and wgrad of every conv class to each other; linearity and the bias-gradient identity cover
the HBM-bound kernels.  Inner products are accumulated in fp64; tolerances are relative to the
product of the operands' norms (fp32 round-off of ~1e6..1e8-term sums)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True, params=["f32", "bf16x6"])
def _conv_math(request):
    """BASELINE-size properties under both conv arithmetics (sae_set_conv_math)."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    hip_lib.set_conv_math(request.param)
    yield request.param
    hip_lib.set_conv_math("f32")


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _close(lhs, rhs, scale, tol=2e-5):
    assert abs(lhs - rhs) <= tol * scale, (lhs, rhs, abs(lhs - rhs) / scale)


@pytest.mark.parametrize("geom", [
    (16, 128, 256, 256, 128, 3, 1, 1),    # D: 3x3 stride 1, the dominant layer (309 GFLOP)
    (16, 128, 257, 257, 256, 3, 2, 0),    # D: 3x3 stride 2 after the blur (transposed-gather dgrad, 2^k+1 grid)
    (16, 128, 256, 256, 256, 1, 1, 0),    # 1x1
    (128, 32, 128, 128, 32, 3, 1, 1),     # Dpatch: narrow layer, batch 128 (wgrad MODE 1)
    (128, 3, 128, 128, 32, 3, 1, 1),      # Dpatch stem (wgrad MODE 2)
    (16, 512, 16, 16, 512, 3, 1, 1),      # tail: split-K path
], ids=str)
def test_conv_adjointness_and_linearity(geom):
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d
    n, c, h, w, m, k, s, p = geom
    g0 = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(n, c, h, w, device=DEV, generator=g0, requires_grad=True)
    wt = torch.randn(m, c, k, k, device=DEV, generator=g0, requires_grad=True)
    alpha = 1.0 / math.sqrt(c * k * k)
    y = conv2d(x, wt, stride=s, padding=p, alpha=alpha)
    gy = torch.randn(y.shape, device=DEV, generator=g0)
    gx, gw = torch.autograd.grad(y, [x, wt], gy)
    lhs = _dot(y, gy)
    scale = float(y.double().norm() * gy.double().norm())
    _close(lhs, _dot(x, gx), scale)          # <conv(x, w), g> = <x, dgrad(g, w)>
    _close(lhs, _dot(wt, gw), scale)         # ... = <w, wgrad(x, g)>  (conv is bilinear)
    x2 = torch.randn(x.shape, device=DEV, generator=g0)
    y2 = conv2d(x2, wt, stride=s, padding=p, alpha=alpha)
    y12 = conv2d(0.5 * x.detach() + x2, wt, stride=s, padding=p, alpha=alpha)
    err = float((y12 - (0.5 * y.detach() + y2.detach())).abs().max())
    assert err <= 2e-5 * float(y12.abs().max()), err


def test_transposed_conv_adjointness():
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv_transpose2d
    g0 = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(16, 256, 128, 128, device=DEV, generator=g0, requires_grad=True)     # G: 256 -> 128, 128 -> 257
    wt = torch.randn(128, 256, 3, 3, device=DEV, generator=g0, requires_grad=True)
    y = conv_transpose2d(x, wt, alpha=1.0 / 48.0)
    assert y.shape == (16, 128, 257, 257)
    gy = torch.randn(y.shape, device=DEV, generator=g0)
    gx, gw = torch.autograd.grad(y, [x, wt], gy)
    lhs = _dot(y, gy)
    scale = float(y.double().norm() * gy.double().norm())
    _close(lhs, _dot(x, gx), scale)
    _close(lhs, _dot(wt, gw), scale)


@pytest.mark.parametrize("pad", [(2, 2), (1, 1)])
def test_blur_adjointness_and_dc_gain(pad):
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import make_kernel
    from swapping_autoencoder_pytorch_amd.stylegan2_op import upfirdn2d
    g0 = torch.Generator(device=DEV).manual_seed(3)
    taps = make_kernel([1, 3, 3, 1]).to(DEV)
    x = torch.randn(16, 128, 256, 256, device=DEV, generator=g0, requires_grad=True)
    y = upfirdn2d(x, taps, pad=pad)
    gy = torch.randn(y.shape, device=DEV, generator=g0)
    gx, = torch.autograd.grad(y, x, gy)
    _close(_dot(y, gy), _dot(x, gx), float(y.double().norm() * gy.double().norm()))
    # taps sum to one: a constant image stays constant in the interior (closed-form KAT)
    const = upfirdn2d(torch.full((2, 4, 256, 256), 3.25, device=DEV), taps, pad=pad)
    interior = const[:, :, 3:-3, 3:-3]
    assert float((interior - 3.25).abs().max()) <= 1e-6


def test_bias_act_identities():
    from swapping_autoencoder_pytorch_amd.stylegan2_op import fused_leaky_relu
    g0 = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(16, 128, 256, 256, device=DEV, generator=g0, requires_grad=True)
    b = torch.randn(128, device=DEV, generator=g0, requires_grad=True)
    y = fused_leaky_relu(x, b)
    ref = torch.nn.functional.leaky_relu(x.detach() + b.detach().view(1, -1, 1, 1), 0.2) * math.sqrt(2)
    assert float((y - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    gy = torch.randn(y.shape, device=DEV, generator=g0)
    gx, gb = torch.autograd.grad(y, [x, b], gy)
    want = gx.double().sum(dim=(0, 2, 3))                       # grad_bias = per-channel sum of grad_input
    assert float((gb.double() - want).abs().max()) <= 1e-5 * float(gx.double().abs().sum() / 128)
    assert float(fused_leaky_relu(torch.zeros(2, 128, 4, 4, device=DEV), b.detach())[0, :, 0, 0].sub(
        torch.nn.functional.leaky_relu(b.detach(), 0.2) * math.sqrt(2)).abs().max()) <= 1e-6   # f(0, b) KAT


def test_upsample_residual_adjointness():
    from swapping_autoencoder_pytorch_amd.stylegan2_op import upsample2x_add
    g0 = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(16, 128, 128, 128, device=DEV, generator=g0, requires_grad=True)
    res = torch.randn(16, 128, 256, 256, device=DEV, generator=g0, requires_grad=True)
    y = upsample2x_add(x, res, 0.7)
    ref = 0.7 * (torch.nn.functional.interpolate(x.detach(), scale_factor=2, mode="bilinear", align_corners=False) + res.detach())
    assert float((y - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    gy = torch.randn(y.shape, device=DEV, generator=g0)
    gx, gres = torch.autograd.grad(y, [x, res], gy)
    scale = float(y.double().norm() * gy.double().norm())
    _close(_dot(y, gy), _dot(x, gx) + _dot(res, gres), scale)
