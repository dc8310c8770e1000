"""Prepared conv weights (include/sae_hip.h "Prepared weights", stylegan2_op/weight_prep.py).

C-ABI: a launch that is handed the layout sae_conv2d_wprep_f32 built gives the SAME BITS as one that re-lays its weights,
never touches the workspace for them, and ignores a buffer offered under a foreign layout identity.
Host cache: a parameter updated through anything autograd sees (in-place ops, state_dict() tensors, FusedAdam) is re-laid;
a write through `.data` is the documented blind spot that `invalidate()` covers; temporaries are never cached.
Runs on the emulated kernels here and on the GPU library with -m gpu."""
import os

import numpy as np
import pytest
import torch

import abi_harness as H

# n, c, h, w, m, k, stride, pad: 3x3 s1 (both tiles), 3x3 s2 forward, its data gradient on the tr2 / tr kernels, 1x1, a thin 1x1
CASES = [(2, 20, 16, 16, 130, 3, 1, 1), (1, 12, 20, 36, 70, 3, 1, 0), (2, 12, 21, 41, 70, 3, 2, 0), (2, 40, 17, 33, 20, 3, 2, 0),
         (1, 24, 35, 35, 70, 3, 2, 0), (1, 70, 16, 16, 70, 1, 1, 0), (1, 3, 40, 40, 24, 1, 1, 0)]


def _check_abi(lib, device):
    rng = np.random.default_rng(3)
    laid_out = 0
    for n, c, h, w, m, k, s, p in CASES:
        d = H.conv_desc(n, c, h, w, m, k, s, p)
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        wt = rng.standard_normal((m, c, k, k)).astype(np.float32)
        gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
        for op, a, shape in ((0, x, gy.shape), (1, gy, x.shape)):
            want = H.conv(lib, op, d, a, wt, shape, alpha=0.37, device=device)
            got, floats = H.conv_prepped(lib, op, d, a, wt, shape, alpha=0.37, device=device)
            assert np.array_equal(want, got), ("prepared weights changed the result", (n, c, h, w, m, k, s, p), op)
            laid_out += int(floats > 0)
            if floats:
                other, _ = H.conv_prepped(lib, op, d, a, wt, shape, alpha=0.37, device=device, wrong_layout=True)
                assert np.array_equal(want, other), ("a foreign layout identity must be ignored", (n, c, h, w, m, k, s, p), op)
    assert laid_out >= 10       # (the thin 1x1 kernels stream the parameter layout: nothing to prepare there)


def test_prepared_weights_abi_on_the_emulator(emu_lib):
    _check_abi(emu_lib, None)


@pytest.mark.gpu
def test_prepared_weights_abi_on_the_gpu():
    from swapping_autoencoder_pytorch_amd import hip_lib
    _check_abi(hip_lib.get(), "cuda:0")


def _check_cache(device):
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import ConvLayer
    from swapping_autoencoder_pytorch_amd.stylegan2_op import weight_prep
    weight_prep.invalidate()
    torch.manual_seed(0)
    layer = ConvLayer(12, 40, 3).to(device)
    x = torch.randn(2, 12, 16, 16, device=device)
    w = layer.Conv.weight if hasattr(layer, "Conv") else next(p for p in layer.parameters() if p.dim() == 4)

    def fresh():          # the same call with the cache off: the launch re-lays the weights itself
        os.environ["SAE_WPREP_CACHE"] = "0"
        try:
            return layer(x).detach().clone()
        finally:
            os.environ.pop("SAE_WPREP_CACHE", None)

    y0 = layer(x).detach().clone()
    n_entries = len(weight_prep._ENTRIES)
    assert n_entries >= 1 and torch.equal(y0, fresh())
    assert torch.equal(layer(x).detach(), y0) and len(weight_prep._ENTRIES) == n_entries       # served from the cache
    with torch.no_grad():
        w.mul_(1.5)                                                    # an update autograd sees
    y1 = layer(x).detach().clone()
    assert not torch.equal(y1, y0) and torch.equal(y1, fresh())
    layer.state_dict()[[k for k in layer.state_dict() if k.endswith("weight")][0]].mul_(0.5)   # state_dict() tensors share the counter
    assert torch.equal(layer(x).detach(), fresh())
    w.data.mul_(2.0)                                                   # the blind spot: .data has its own version counter
    stale = layer(x).detach().clone()
    weight_prep.invalidate()
    assert torch.equal(layer(x).detach(), fresh()) and not torch.equal(stale, fresh())
    # a temporary tensor in the weight position is never cached
    from swapping_autoencoder_pytorch_amd.stylegan2_op.conv2d_gemm import _Geom, _fwd
    before = len(weight_prep._ENTRIES)
    _fwd(x, torch.randn(40, 12, 3, 3, device=device), _Geom(2, 12, 16, 16, 40, 3, 1, 1, False, 1.0))
    assert len(weight_prep._ENTRIES) == before
    # the gradient path (dgrad layout) and two optimiser steps through FusedAdam, which bumps the versions itself
    from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
    opt = FusedAdam(layer.parameters(), lr=0.01, betas=(0.0, 0.99))
    xg = x.clone().requires_grad_()
    for _ in range(2):
        opt.zero_grad()
        v = w._version
        layer(xg).square().mean().backward()
        opt.step()
        assert w._version > v
        assert torch.equal(layer(x).detach(), fresh())
    # a parameter that dies takes its prepared copies (and their device buffers) with it
    import gc
    assert len(weight_prep._ENTRIES) >= 1
    del layer, w, opt, xg
    gc.collect()
    assert len(weight_prep._ENTRIES) == 0


def test_prepared_weights_cache_follows_the_version_counter_on_the_emulator(emu_lib):
    import parity_common as P
    with P.backend(emu_lib):
        _check_cache("cpu")


@pytest.mark.gpu
def test_prepared_weights_cache_follows_the_version_counter_on_the_gpu():
    _check_cache("cuda:0")
