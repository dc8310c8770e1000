"""csrc/glue.hip: util.normalize, GeneratorModulation and gan_loss as single kernels.  The oracle is pinned to the ATen
expression the reference writes at each call site (util/util.py:18-22, generator.py:62-67, loss.py:10-16) incl. its
autograd; the emulated kernels (CPU) and the real kernels (``-m gpu``) to the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import abi_harness as H

NORM_SHAPES = [(16, 8, 16, 16), (16, 2048), (3, 5, 7, 3), (2, 70), (4, 1, 9, 9)]
AFFINE_SHAPES = [(16, 8, 16, 16), (3, 5, 7, 3), (2, 4, 1, 1)]
LOSS_SHAPES = [(16, 1), (5, 7), (128, 1)]


def _run(lib, oracle_lib, device):
    rng = np.random.default_rng(8)
    for shape in NORM_SHAPES:
        x = (rng.standard_normal(shape) * 3).astype(np.float32)
        gy = rng.standard_normal(shape).astype(np.float32)
        assert np.allclose(H.l2_normalize(lib, x, device=device), H.l2_normalize(oracle_lib, x), rtol=5e-7, atol=1e-9)
        a, o = H.l2_normalize_bwd(lib, gy, x, device=device), H.l2_normalize_bwd(oracle_lib, gy, x)
        # gx = r gy - r^3 <gy, x> x: two terms of size r |gy| that cancel (completely when C = 1): the error scale is
        # that of the terms, not of their difference
        r = 1.0 / np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True) + 1e-8)
        assert np.abs(a - o).max() <= 3e-6 * float((np.abs(gy) * r).max())
    for shape in AFFINE_SHAPES:
        x = rng.standard_normal(shape).astype(np.float32)
        a = rng.standard_normal(shape[:2]).astype(np.float32)
        b = rng.standard_normal(shape[:2]).astype(np.float32)
        g = rng.standard_normal(shape).astype(np.float32)
        assert np.allclose(H.plane_affine(lib, x, a, b, device=device), H.plane_affine(oracle_lib, x, a, b), rtol=3e-7, atol=1e-7)
        r, o = H.plane_affine_bwd(lib, g, x, a, device=device), H.plane_affine_bwd(oracle_lib, g, x, a)
        assert np.array_equal(r[0], o[0])
        hw = int(np.prod(shape[2:]))
        assert np.abs(r[1] - o[1]).max() <= 1e-6 * hw and np.abs(r[2] - o[2]).max() <= 1e-6 * hw
    for shape in LOSS_SHAPES:
        x = (rng.standard_normal(shape) * 4).astype(np.float32)
        x.flat[0] = 25.0            # beyond F.softplus's threshold
        gy = rng.standard_normal(shape[0]).astype(np.float32)
        for sign in (1.0, -1.0):
            assert np.allclose(H.softplus_mean(lib, x, sign, device=device), H.softplus_mean(oracle_lib, x, sign), rtol=1e-6, atol=1e-7)
            assert np.allclose(H.softplus_mean_bwd(lib, gy, x, sign, device=device), H.softplus_mean_bwd(oracle_lib, gy, x, sign),
                               rtol=2e-6, atol=1e-8)


def test_oracle_is_the_reference_expression(oracle_lib):
    rng = np.random.default_rng(9)
    for shape in NORM_SHAPES:
        x = torch.from_numpy((rng.standard_normal(shape) * 3).astype(np.float32)).double().requires_grad_()
        gy = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).double()
        y = x * torch.rsqrt(torch.sum(x ** 2, dim=1, keepdim=True) + 1e-8)          # util/util.py:18-22
        gx, = torch.autograd.grad(y, x, gy)
        xn, gn = x.detach().float().numpy(), gy.float().numpy()
        assert np.allclose(H.l2_normalize(oracle_lib, xn), y.detach().numpy(), rtol=2e-7, atol=1e-12)
        assert np.allclose(H.l2_normalize_bwd(oracle_lib, gn, xn), gx.numpy(), rtol=1e-6, atol=1e-9)
    for shape in AFFINE_SHAPES:
        x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).double().requires_grad_()
        a = torch.from_numpy(rng.standard_normal(shape[:2]).astype(np.float32)).double().requires_grad_()
        b = torch.from_numpy(rng.standard_normal(shape[:2]).astype(np.float32)).double().requires_grad_()
        g = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).double()
        y = x * (1 * a[:, :, None, None]) + b[:, :, None, None]                       # generator.py:62-67
        gx, ga, gb = torch.autograd.grad(y, [x, a, b], g)
        f = lambda t: t.detach().float().numpy()
        assert np.allclose(H.plane_affine(oracle_lib, f(x), f(a), f(b)), y.detach().numpy(), rtol=2e-7, atol=1e-12)
        r = H.plane_affine_bwd(oracle_lib, f(g), f(x), f(a))
        for got, want in zip(r, (gx, ga, gb)):
            assert np.allclose(got, want.numpy(), rtol=1e-6, atol=1e-6)
    for shape in LOSS_SHAPES:
        x = torch.from_numpy((rng.standard_normal(shape) * 4).astype(np.float32)).double().requires_grad_()
        gy = torch.from_numpy(rng.standard_normal(shape[0]).astype(np.float32)).double()
        for sign in (1.0, -1.0):
            y = F.softplus(sign * x).view(x.size(0), -1).mean(dim=1)                  # loss.py:10-16
            gx, = torch.autograd.grad(y, x, gy)
            assert np.allclose(H.softplus_mean(oracle_lib, x.detach().float().numpy(), sign), y.detach().numpy(), rtol=2e-7, atol=1e-9)
            assert np.allclose(H.softplus_mean_bwd(oracle_lib, gy.float().numpy(), x.detach().float().numpy(), sign), gx.numpy(),
                               rtol=1e-6, atol=1e-9)


def test_emulated_kernels_vs_oracle(emu_lib, oracle_lib):
    _run(emu_lib, oracle_lib, None)


@pytest.mark.gpu
def test_gpu_kernels_vs_oracle(oracle_lib):
    from swapping_autoencoder_pytorch_amd import hip_lib
    _run(hip_lib.get(), oracle_lib, "cuda:0")
