"""The package's whole encoder and generator (host modules over the C-ABI, oracle back end) against the ATen restatement of the
reference's modules (oracle/aten_cpu_path.py, pinned to the reference in tests/dropin_ref_worker.py::aten_cpu_path_pin): one
reconstruction, output and every parameter gradient, fixed noise maps.  The GPU form of this test at BASELINE size is
tests/test_gpu_network_parity.py."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from parity_common import backend


@pytest.mark.parametrize("n_sp", [2, 3])     # 3: an upsampling block with an Identity skip (generator.py:48-49)
def test_encoder_generator_reconstruction_matches_the_aten_restatement(oracle_lib, n_sp):
    import aten_cpu_path as A
    from swapping_autoencoder_pytorch_amd.networks.encoder import StyleGAN2ResnetEncoder
    from swapping_autoencoder_pytorch_amd.networks.generator import StyleGAN2ResnetGenerator
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import NoiseInjection
    opt = make_options("tiny32", batch_size=2, num_gpus=0, netE_scale_capacity=0.25, netG_scale_capacity=0.125,
                       global_code_ch=64, spatial_code_ch=8, netE_num_downsampling_sp=n_sp,
                       netE_num_downsampling_gl=4 - n_sp)
    torch.manual_seed(0)
    with backend(oracle_lib):
        enc, gen = StyleGAN2ResnetEncoder(opt), StyleGAN2ResnetGenerator(opt)
        ref_enc, ref_gen = A.EncoderCPU(opt), A.GeneratorCPU(opt)
        sp = list(ref_enc.parameters()) + list(ref_gen.parameters())
        dp = list(enc.parameters()) + list(gen.parameters())
        assert [tuple(p.shape) for p in sp] == [tuple(p.shape) for p in dp]
        with torch.no_grad():
            for a, b in zip(sp, dp):
                v = torch.randn(a.shape) * (1.0 if a.dim() > 1 and tuple(a.shape) != (1, 3, 1, 1) else 0.2)
                a.copy_(v)
                b.copy_(v)
        x = torch.rand(2, 3, 32, 32) * 2 - 1
        with torch.no_grad():
            gen(*enc(x[:1]))
        mine = [m for m in gen.modules() if isinstance(m, NoiseInjection)]
        theirs = [m for m in ref_gen.modules() if isinstance(m, A.StyledConvCPU)]
        assert len(mine) == len(theirs) == 2 * (opt.netG_num_base_resnet_layers + opt.netE_num_downsampling_sp)
        for m, r in zip(mine, theirs):
            z = torch.randn(2, 1, m.image_size[2], m.image_size[3])
            m.fixed_noise = z
            r.fixed_noise = z
        y, yr = gen(*enc(x)), ref_gen(*ref_enc(x))
        assert float((y - yr).abs().max() / yr.abs().max()) < 1e-5
        g1 = torch.autograd.grad((y - x).abs().mean(), dp)
        g2 = torch.autograd.grad((yr - x).abs().mean(), sp)
        for a, b in zip(g1, g2):
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max() + 1e-12)


def test_aten_cpu_train_iteration_reproduces_the_reference_golden_steps(oracle_lib):
    """oracle/aten_cpu_path.TrainIterationCPU (bench.py --full-cpu-baseline: one discriminator call + one generator call of
    the reference's driver on its CPU path, crops through F.grid_sample, torch.optim.Adam) against the loss dictionaries the
    REFERENCE produced on the same weights, images and random stream (tests/golden: micro_steps step0 / step1 -- the second
    one after the first one's Adam update)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import aten_cpu_path as A
    import parity_common as P
    from param_recipe import uniform_images
    with P.backend(oracle_lib):
        opt, model, net = P.build_micro("cpu")           # the mirror model only lends its (recipe-filled) parameters
    it = A.TrainIterationCPU(opt)
    for name, m in it.modules().items():
        src, dst = list(getattr(net, name).parameters()), list(m.parameters())
        assert [tuple(a.shape) for a in src] == [tuple(b.shape) for b in dst], name
        with torch.no_grad():
            for a, b in zip(src, dst):
                b.copy_(a)
    _, info = P.golden()
    for call, fn in ((0, it.discriminator_call), (1, it.generator_call)):
        torch.manual_seed(1000 + call)
        got = fn(uniform_images(4, 32, 600 + call))
        want = info["micro_steps"]["step%d" % call]
        for k, v in got.items():
            assert abs(v - want[k]) <= 2e-6 * max(1.0, abs(want[k])), (call, k, v, want[k])


def test_mask_frozen_double_reference_recovers_this_packages_sign_patterns(oracle_lib):
    """tests/mask_frozen.py on the oracle back end: every activation call of the double restatement finds the tensor this
    package saved for the same activation (so the GPU test's frozen reference has no silent holes), the double run under
    those sign patterns is within the north-star tolerance of this package's gradients, and a recorded restatement run
    replays onto itself with zero flips."""
    import aten_cpu_path as A
    from mask_frozen import SavedActivations, frozen_reference_grads
    from swapping_autoencoder_pytorch_amd.networks.encoder import StyleGAN2ResnetEncoder
    from swapping_autoencoder_pytorch_amd.networks.generator import StyleGAN2ResnetGenerator
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import NoiseInjection
    opt = make_options("tiny32", batch_size=2, num_gpus=0, netE_scale_capacity=0.25, netG_scale_capacity=0.125,
                       global_code_ch=64, spatial_code_ch=8, netE_num_downsampling_sp=2, netE_num_downsampling_gl=2)
    torch.manual_seed(0)
    with backend(oracle_lib):
        enc, gen = StyleGAN2ResnetEncoder(opt), StyleGAN2ResnetGenerator(opt)
        ref_enc, ref_gen = A.EncoderCPU(opt), A.GeneratorCPU(opt)
        sp = list(ref_enc.parameters()) + list(ref_gen.parameters())
        dp = list(enc.parameters()) + list(gen.parameters())
        with torch.no_grad():
            for a, b in zip(sp, dp):
                v = torch.randn(a.shape) * (1.0 if a.dim() > 1 and tuple(a.shape) != (1, 3, 1, 1) else 0.2)
                a.copy_(v)
                b.copy_(v)
        x = torch.rand(2, 3, 32, 32) * 2 - 1
        with torch.no_grad():
            gen(*enc(x[:1]))
        mine = [m for m in gen.modules() if isinstance(m, NoiseInjection)]
        theirs = [m for m in ref_gen.modules() if isinstance(m, A.StyledConvCPU)]
        for m, r in zip(mine, theirs):
            z = torch.randn(2, 1, m.image_size[2], m.image_size[3])
            m.fixed_noise = z
            r.fixed_noise = z.double()
        t = torch.randn(2, 3, 32, 32)
        xo = x.clone().requires_grad_(True)
        with SavedActivations() as saved:
            y = gen(*enc(xo))
        g_ours = torch.autograd.grad((y * t).sum(), [xo] + dp)

        class Both(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.E, self.G = ref_enc, ref_gen
        both = Both().double()
        run = lambda m, i: m.G(*m.E(i[0]))
        y64, g64, flips = frozen_reference_grads(A, both, [x], run, t, saved=saved)
        assert saved.unmatched == [] and saved.matched == len(flips) > 10, (saved.unmatched, saved.matched)
        assert float((y - y64).abs().max() / y64.abs().max()) < 1e-5
        for a, b in zip(g_ours, g64):
            assert float((a.double() - b).norm()) <= 1e-4 * float(b.norm() + 1e-30)
        with A.ActivationMasks.record() as rec:
            run(both, [x.double()])
        _, g_again, flips2 = frozen_reference_grads(A, both, [x], run, t, masks=rec.masks)
        assert sum(flips2) == 0 and len(flips2) == len(flips)
