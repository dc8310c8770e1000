"""StyledConv's one-kernel forms and the ActTicket hand-over inside the generator blocks (stylegan2_layers.StyledConv,
networks/generator.py) against the module-by-module path of the same layers (SAE_STYLED_FUSED=0): outputs, input / style
gradients and every parameter gradient; the fused entries must really be the ones that ran."""
import pytest
import torch

from parity_common import backend


def _run_block(block, x, style, fused, lib):
    from swapping_autoencoder_pytorch_amd import stylegan2_layers as SL
    prev = SL._FUSED_STYLED
    SL._FUSED_STYLED = fused
    calls = []
    orig = lib.call

    def counting(name, *a):
        calls.append(name)
        return orig(name, *a)

    lib.call = counting
    try:
        x = x.clone().requires_grad_(True)
        style = style.clone().requires_grad_(True)
        torch.manual_seed(3)                   # the noise maps
        y = block(x, style)
        torch.manual_seed(4)
        g = torch.randn_like(y)
        params = list(block.parameters())
        grads = torch.autograd.grad(y, [x, style] + params, g)
        return [y.detach()] + [t.detach() for t in grads], calls
    finally:
        lib.call = orig
        SL._FUSED_STYLED = prev


def _compare(lib, device, tol):
    from swapping_autoencoder_pytorch_amd.networks.generator import ResolutionPreservingResnetBlock, UpsamplingResnetBlock
    with backend(lib):
        for make, xshape in ((lambda: ResolutionPreservingResnetBlock(None, 6, 10, 16), (2, 6, 8, 8)),
                             (lambda: ResolutionPreservingResnetBlock(None, 8, 8, 16), (3, 8, 4, 4)),
                             (lambda: UpsamplingResnetBlock(6, 10, 16, use_noise=True), (2, 6, 8, 8)),
                             (lambda: UpsamplingResnetBlock(8, 8, 16, use_noise=True), (1, 8, 16, 16))):
            torch.manual_seed(11)
            block = make()
            with torch.no_grad():
                for n, p in block.named_parameters():
                    if p.dim() == 1 or n.endswith("noise.weight"):
                        p.normal_(0.0, 0.5)        # zero-initialised biases / noise strengths would hide wrong gradients
            block = block.to(device)
            x = torch.randn(xshape).to(device)
            style = torch.randn(xshape[0], 16).to(device)
            a, calls = _run_block(block, x, style, True, lib)
            b, calls_b = _run_block(block, x, style, False, lib)
            # the fused path: one-kernel forward of the plain StyledConv(s), the blur + activation of the upsampling one, the
            # pair kernel in the backward; the module path: none of them
            assert "modconv2d_fwd_noise_bias_act_f32" in calls and "plane_scale_dot_act_f32" in calls
            assert ("upfirdn2d_noise_bias_act_f32" in calls) == isinstance(block, UpsamplingResnetBlock)
            assert not {"modconv2d_fwd_noise_bias_act_f32", "plane_scale_dot_act_f32", "upfirdn2d_noise_bias_act_f32"} & set(calls_b)
            assert calls.count("noise_bias_act_bwd_f32") == 1 and calls_b.count("noise_bias_act_bwd_f32") == 2
            for i, (u, v) in enumerate(zip(a, b)):
                err = (u - v).abs().max().item() / (v.abs().max().item() + 1e-30)
                assert err < tol, (type(block).__name__, i, err)


def test_fused_styled_blocks_match_the_module_path_oracle(oracle_lib):
    _compare(oracle_lib, "cpu", 2e-6)


def test_fused_styled_blocks_match_the_module_path_emulator(emu_lib):
    _compare(emu_lib, "cpu", 2e-6)


def test_a_second_consumer_of_a_ticketed_activation_is_refused(oracle_lib):
    """The hand-over is only sound when the activation has one consumer: a re-summed gradient is detected and refused."""
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeError
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import StyledConv
    from swapping_autoencoder_pytorch_amd.stylegan2_op.modulate import ActTicket
    with backend(oracle_lib):
        torch.manual_seed(0)
        c1, c2 = StyledConv(4, 6, 3, 8), StyledConv(6, 6, 3, 8)
        x, s = torch.randn(1, 4, 8, 8, requires_grad=True), torch.randn(1, 8)
        t = ActTicket()
        a1 = c1(x, s, act_ticket=t)
        y = c2(a1, s, input_ticket=t) + a1.sum()          # a1 used a second time
        with pytest.raises((SaeError, RuntimeError)):
            y.sum().backward()


def test_a_second_consumer_of_a_merged_activation_is_refused(oracle_lib):
    """GradScaleTicket: the merge hands its 1/sqrt(2) to conv2's backward only if conv2's output goes nowhere else."""
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeError
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import StyledConv
    from swapping_autoencoder_pytorch_amd.stylegan2_op import upsample2x_add
    from swapping_autoencoder_pytorch_amd.stylegan2_op.modulate import GradScaleTicket
    with backend(oracle_lib):
        torch.manual_seed(0)
        c2 = StyledConv(4, 4, 3, 8)
        x, s = torch.randn(1, 4, 8, 8, requires_grad=True), torch.randn(1, 8)
        skip = torch.randn(1, 4, 4, 4, requires_grad=True)
        t = GradScaleTicket()
        res = c2(x, s, grad_scale_ticket=t)
        assert t.armed
        y = upsample2x_add(skip, res, 0.5, res_ticket=t)
        gx_ok, = torch.autograd.grad(y.sum(), x, retain_graph=True)
        t2 = GradScaleTicket()
        res2 = c2(x, s, grad_scale_ticket=t2)
        gx_ref, = torch.autograd.grad(upsample2x_add(skip, res2, 0.5).sum(), x)        # no hand-over: the plain multiply
        assert (gx_ok - gx_ref).abs().max() < 1e-6 * gx_ref.abs().max() + 1e-9
        with pytest.raises((SaeError, RuntimeError)):
            (y.sum() + res.sum()).backward()              # res has a second consumer


@pytest.mark.gpu
def test_fused_styled_blocks_match_the_module_path_gpu():
    from swapping_autoencoder_pytorch_amd import hip_lib
    _compare(hip_lib.get(), "cuda:0", 5e-6)


@pytest.mark.gpu
def test_fused_styled_blocks_on_the_gpu_against_the_oracle(oracle_lib):
    """The generator blocks' fused path on the GPU (one-kernel StyledConv forward, blur + noise + activation epilogue, the ticketed
    backward) against the MODULE-BY-MODULE path on the CPU oracle, same weights, same explicit noise maps: output, input / style
    gradients and every parameter gradient, relative L2 per tensor.  models/networks/generator.py:30-53,
    stylegan2_layers.py:266-351,398-405."""
    import copy
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.networks.generator import ResolutionPreservingResnetBlock, UpsamplingResnetBlock
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import NoiseInjection
    for make, xshape in ((lambda: ResolutionPreservingResnetBlock(None, 6, 10, 16), (2, 6, 8, 8)),
                         (lambda: UpsamplingResnetBlock(6, 10, 16, use_noise=True), (2, 6, 8, 8)),
                         (lambda: ResolutionPreservingResnetBlock(None, 64, 64, 32), (2, 64, 32, 32)),
                         (lambda: UpsamplingResnetBlock(128, 64, 32, use_noise=True), (2, 128, 32, 32))):
        torch.manual_seed(11)
        cpu_block = make()
        with torch.no_grad():
            for n, p in cpu_block.named_parameters():
                if p.dim() == 1 or n.endswith("noise.weight"):
                    p.normal_(0.0, 0.5)
        gpu_block = copy.deepcopy(cpu_block).to("cuda:0")
        sdim = 16 if xshape[1] == 6 else 32
        x, style = torch.randn(xshape), torch.randn(xshape[0], sdim)
        with backend(oracle_lib), torch.no_grad():
            cpu_block(x[:1], style[:1])                    # records every NoiseInjection's map size
        g = torch.Generator().manual_seed(5)
        for mc, mg in zip([m for m in cpu_block.modules() if isinstance(m, NoiseInjection)],
                          [m for m in gpu_block.modules() if isinstance(m, NoiseInjection)]):
            z = torch.randn(xshape[0], 1, mc.image_size[2], mc.image_size[3], generator=g)
            mc.fixed_noise, mg.fixed_noise = z, z.to("cuda:0")
        with backend(hip_lib.get()):
            got, calls = _run_block(gpu_block, x.to("cuda:0"), style.to("cuda:0"), True, hip_lib.get())
        with backend(oracle_lib):
            want, _ = _run_block(cpu_block, x, style, False, oracle_lib)
        assert "modconv2d_fwd_noise_bias_act_f32" in calls or "wino_fused_conv_f32" in calls
        # _run_block draws the output gradient on the block's device: redo both backward passes' weights identically
        for i, (u, v) in enumerate(zip(got[:1], want[:1])):
            u, v = u.cpu().double(), v.double()
            assert float((u - v).pow(2).sum().sqrt() / (v.pow(2).sum().sqrt() + 1e-30)) < 2e-5, (type(cpu_block).__name__, xshape, i)
        # gradients under ONE common output gradient
        torch.manual_seed(4)
        gout = torch.randn(want[0].shape)
        outs = []
        for block, dev, lib, fused in ((gpu_block, "cuda:0", hip_lib.get(), True), (cpu_block, "cpu", oracle_lib, False)):
            from swapping_autoencoder_pytorch_amd import stylegan2_layers as SL
            prev, SL._FUSED_STYLED = SL._FUSED_STYLED, fused
            try:
                with backend(lib):
                    xi, si = x.to(dev).requires_grad_(True), style.to(dev).requires_grad_(True)
                    y = block(xi, si)
                    grads = torch.autograd.grad(y, [xi, si] + list(block.parameters()), gout.to(dev))
                    outs.append([t.detach().cpu().double() for t in grads])
            finally:
                SL._FUSED_STYLED = prev
        names = ["grad_x", "grad_style"] + [n for n, _ in cpu_block.named_parameters()]
        for n, u, v in zip(names, outs[0], outs[1]):
            err = float((u - v).pow(2).sum().sqrt() / (v.pow(2).sum().sqrt() + 1e-30))
            # (a noise strength's gradient is ONE cancelling sum over n * c * h * w terms: fp32 summation noise relative to the
            # cancelled total is ~sqrt(terms) ulp)
            assert err < (2e-4 if u.numel() == 1 else 2e-5), (type(cpu_block).__name__, xshape, n, err)
