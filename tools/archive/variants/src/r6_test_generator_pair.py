"""StyleGAN2ResnetGenerator.forward_pair (networks/generator.py): the reconstruction pass and the hybrid pass of a training call
(swapping_autoencoder_model.py:122-124,192-201 of the reference) as ONE pass over the concatenated batch must give what the two
separate passes give -- outputs and every parameter / input gradient -- with the SAME random stream: the noise maps are drawn
ahead in the order the two passes would draw them."""
import pytest
import torch

from parity_common import backend


def _tiny_generator():
    from swapping_autoencoder_pytorch_amd.networks.generator import StyleGAN2ResnetGenerator
    from swapping_autoencoder_pytorch_amd.options import make_options
    opt = make_options("tiny32", batch_size=4, num_gpus=0, netG_scale_capacity=0.125, global_code_ch=32, spatial_code_ch=4,
                       netE_num_downsampling_sp=2)
    torch.manual_seed(0)
    g = StyleGAN2ResnetGenerator(opt)
    with torch.no_grad():
        for n, p in g.named_parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.3)          # zero-initialised biases and noise strengths would hide the noise maps
    return g


def _run(g, pair, device):
    torch.manual_seed(7)
    sp = torch.randn(4, 4, 8, 8).to(device).requires_grad_(True)
    gl = torch.randn(4, 32).to(device).requires_grad_(True)
    torch.manual_seed(11)                        # the noise stream
    if pair:
        a, b = g.forward_pair(sp[:2], gl[:2], sp.flip(0), gl)
    else:
        a, b = g(sp[:2], gl[:2]), g(sp.flip(0), gl)
    torch.manual_seed(13)
    loss = (a * torch.randn(a.shape).to(device)).sum() + (b * torch.randn(b.shape).to(device)).sum()
    grads = torch.autograd.grad(loss, [sp, gl] + list(g.parameters()))
    return [a.detach(), b.detach()] + [t.detach() for t in grads]


def _compare(lib, device, tol):
    with backend(lib):
        g = _tiny_generator().to(device)
        _run(g, False, device)                   # the first pass of a network records the noise-map sizes
        two = _run(g, False, device)
        one = _run(g, True, device)
        assert len(one) == len(two)
        for i, (u, v) in enumerate(zip(one, two)):
            err = float((u - v).abs().max() / (v.abs().max() + 1e-30))
            assert err < tol, (i, err)
        # the batched pass left nothing queued, and an odd call still falls back to two passes
        from swapping_autoencoder_pytorch_amd.stylegan2_layers import NoiseInjection
        assert all(not m.queued for m in g.modules() if isinstance(m, NoiseInjection))


def test_one_batched_pass_equals_two_passes_on_the_oracle(oracle_lib):
    _compare(oracle_lib, "cpu", 5e-6)


@pytest.mark.gpu
def test_one_batched_pass_equals_two_passes_on_the_gpu():
    from swapping_autoencoder_pytorch_amd import hip_lib
    _compare(hip_lib.get(), "cuda:0", 2e-5)
