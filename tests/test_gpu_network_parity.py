"""Network-level parity at BASELINE size on the GPU (church256, B = 16): the image discriminator (16 x 3 x 256 x 256 ->
logits, stylegan2_layers.py:696-763) and one upsampling block of the generator (16 x 256 x 128 x 128 -> 16 x 128 x 256 x
256 with explicit noise maps, generator.py:39-53) of THIS package against the ATen restatement of the reference's code path
(oracle/aten_cpu_path.py, pinned to the reference's own modules in tests/test_dropin_train.py::
test_aten_cpu_path_matches_reference_discriminator) moved to cuda:0 and run in DOUBLE precision — same weights, same input:
the outputs within 1e-4 of the tensor's largest magnitude, the north-star tolerance, under both conv arithmetics -- for the outputs.  For gradients that tolerance is not attainable by any fp32 implementation of a
network this deep (see _check): the same restatement in fp32 through ATen / MIOpen rides along as the control, and every
gradient tensor must be no further from the double run than 3x what stock fp32 PyTorch-ROCm is.  The kernel-level full-size cases (test_gpu_fullsize_oracle.py) check each conv class against the
double-accumulating oracle; this checks that the layers are wired, scaled and accumulated the same through a whole network
at the real size.  Observed errors are appended to gpurun_out/network_parity.jsonl when that directory exists."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _log(record):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "network_parity.jsonl"), "a") as f:
            f.write(json.dumps(record) + "\n")


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def _stats(a, b):
    """(max-norm error, relative L2 error, fraction of elements further than TOL * max|b| from b)"""
    a, b = a.detach().double(), b.detach().double()
    d = (a - b).abs()
    scale = float(b.abs().max()) + 1e-30
    return (float(d.max()) / scale, float(d.pow(2).sum().sqrt() / (b.pow(2).sum().sqrt() + 1e-30)),
            float((d > TOL * scale).double().mean()))


def _copy_params(src, dst, seed):
    sp, dp = list(src.parameters()), list(dst.parameters())
    assert [tuple(p.shape) for p in sp] == [tuple(p.shape) for p in dp]
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for a, b in zip(sp, dp):
            v = torch.randn(a.shape, generator=g) * (1.0 if a.dim() > 1 else 0.2)
            a.copy_(v)
            b.copy_(v)
    return sp, dp


@pytest.fixture(params=["f32", "bf16x6"])
def conv_math(request):
    from swapping_autoencoder_pytorch_amd import hip_lib
    lib = hip_lib.get()
    lib.call("set_conv_math", hip_lib.CONV_MATH_MODES[request.param])
    try:
        yield request.param
    finally:
        lib.call("set_conv_math", 0)


def _grads_and_errors(ref64, ref32, ours, inputs, run_ref, run_ours, loss_weight):
    """(errors of ours vs the double run, errors of ATen fp32 vs the double run): dicts name -> relative error"""
    r64 = [t.detach().double().requires_grad_(True) for t in inputs]
    r32 = [t.detach().clone().requires_grad_(True) for t in inputs]
    mine = [t.detach().clone().requires_grad_(True) for t in inputs]
    y64, y32, yo = run_ref(ref64, r64), run_ref(ref32, r32), run_ours(ours, mine)
    g64 = torch.autograd.grad((y64 * loss_weight.double()).sum(), r64 + list(ref64.parameters()))
    g32 = torch.autograd.grad((y32 * loss_weight).sum(), r32 + list(ref32.parameters()))
    go = torch.autograd.grad((yo * loss_weight).sum(), mine + list(ours.parameters()))
    names = ["grad_input%d" % i for i in range(len(inputs))] + ["grad " + n for n, _ in ours.named_parameters()]
    err_o, err_a = {"output": _stats(yo, y64)}, {"output": _stats(y32, y64)}
    pooled = {}
    for n, a, b, c in zip(names, go, g32, g64):
        if c.numel() == 1:
            # one-element gradients (the NoiseInjection strengths: ONE cancelling sum over N * C * H * W terms each) are
            # compared as a vector per parameter kind: the error of a single such sum is one random draw, and the ratio of
            # two draws (ours / the control's) exceeds any fixed factor now and then
            kind = "grad *." + ".".join(n.split(".")[-2:]) + " (pooled one-element gradients)"
            pooled.setdefault(kind, ([], [], []))
            for lst, t in zip(pooled[kind], (a, b, c)):
                lst.append(t.detach().reshape(1).double())
            continue
        err_o[n], err_a[n] = _stats(a, c), _stats(b, c)
    for kind, (la, lb, lc) in pooled.items():
        a, b, c = torch.cat(la), torch.cat(lb), torch.cat(lc)
        err_o[kind], err_a[kind] = _stats(a, c), _stats(b, c)
    return err_o, err_a


def _check(case, conv_math, err_o, err_a):
    """What can be asserted at this depth (measured, profiles/r3_network_parity.jsonl): the network OUTPUT is continuous in
    its inputs and agrees with the double run to 3e-6.  GRADIENTS do not: leaky-ReLU masks are discontinuous, a
    pre-activation within an ulp of zero takes the other branch in fp32 than in double, and the sums behind bias / weight
    gradients cancel heavily, so ANY fp32 implementation of this network is 0.5e-3 ... 1e-3 (L2) and up to 5e-2 (max norm,
    input gradient) away from the double run -- stock ATen / MIOpen fp32 shows the same figures as this package, tensor by
    tensor.  The gradient check is therefore relative to that control: no gradient tensor may be further from the double
    run (L2) than 3x what stock fp32 PyTorch-ROCm is on the same tensor (floor 1e-4, the north-star tolerance)."""
    ratio = {k: err_o[k][1] / max(err_a[k][1], 1e-30) for k in err_o}
    worst = max(err_o, key=lambda k: ratio[k] if err_o[k][1] > TOL else 0.0)
    _log({"case": case, "conv_math": conv_math, "tensors": len(err_o), "output_max_err": err_o["output"][0],
          "max over tensors (max norm, l2)": [max(v[i] for v in err_o.values()) for i in range(2)],
          "aten_fp32_control max over tensors (max norm, l2)": [max(v[i] for v in err_a.values()) for i in range(2)],
          "worst tensor relative to the control": worst, "its l2 error": err_o[worst][1], "control's l2 error": err_a[worst][1]})
    assert err_o["output"][0] < TOL, err_o["output"]
    bad = {k: (v[1], err_a[k][1]) for k, v in err_o.items() if not v[1] <= max(3.0 * err_a[k][1], TOL)}
    assert not bad, bad


def test_discriminator_church256_b16_vs_aten_restatement(conv_math):
    import aten_cpu_path as A
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import Discriminator
    ref32 = A.DiscriminatorCPU(256, 2).to(DEV)
    ours = Discriminator(256, 2).to(DEV)
    _copy_params(ref32, ours, 7)
    ref64 = A.DiscriminatorCPU(256, 2).to(DEV).double()
    ref64.load_state_dict({k: v.double() for k, v in ref32.state_dict().items()})
    torch.manual_seed(1)
    x = (torch.rand(16, 3, 256, 256) * 2 - 1).to(DEV)
    w = torch.linspace(-1.0, 1.0, 16, device=DEV).view(16, 1)          # a loss with both signs
    err_o, err_a = _grads_and_errors(ref64, ref32, ours, [x], lambda m, i: m(i[0]), lambda m, i: m(i[0]), w)
    _check("Discriminator 16x3x256x256 vs ATen restatement in double (cuda:0)", conv_math, err_o, err_a)


def test_generator_upsampling_block_128_to_256_vs_aten_restatement(conv_math):
    import aten_cpu_path as A
    from swapping_autoencoder_pytorch_amd.networks.generator import UpsamplingResnetBlock
    inch, outch, styledim, b = 256, 128, 2048, 16          # church256: the last upsampling block of G, global code 2048
    ref32 = A.UpsamplingResnetBlockCPU(inch, outch, styledim).to(DEV)
    ours = UpsamplingResnetBlock(inch, outch, styledim, use_noise=True).to(DEV)
    _copy_params(ref32, ours, 9)
    ref64 = A.UpsamplingResnetBlockCPU(inch, outch, styledim).to(DEV).double()
    ref64.load_state_dict({k: v.double() for k, v in ref32.state_dict().items()})
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, inch, 128, 128, generator=g).to(DEV)
    style = torch.randn(b, styledim, generator=g).to(DEV)
    z1 = torch.randn(b, 1, 256, 256, generator=g).to(DEV)
    z2 = torch.randn(b, 1, 256, 256, generator=g).to(DEV)
    ours.conv1.noise.fixed_noise, ours.conv2.noise.fixed_noise = z1, z2
    t = torch.randn(b, outch, 256, 256, generator=g).to(DEV)

    def run_ref(m, i):
        return m(i[0], i[1], z1.to(i[0].dtype), z2.to(i[0].dtype))

    err_o, err_a = _grads_and_errors(ref64, ref32, ours, [x, style], run_ref, lambda m, i: m(i[0], i[1]), t)
    _check("UpsamplingResnetBlock 16x256x128x128 -> 16x128x256x256 vs ATen restatement in double (cuda:0)", conv_math, err_o, err_a)


class _Reconstruction(torch.nn.Module):
    """image -> E -> (spatial code, global code) -> G -> image: the reconstruction path of every training step
    (swapping_autoencoder_model.py:96-101)"""

    def __init__(self, enc, gen):
        super().__init__()
        self.E, self.G = enc, gen

    def forward(self, x):
        sp, gl = self.E(x)
        return self.G(sp, gl)


def test_encoder_generator_reconstruction_church256_b16_vs_aten_restatement(conv_math):
    """The whole encoder and the whole generator at the BASELINE configuration (church256, B = 16, default widths): one
    reconstruction, gradients of the input image and of all 100 parameter tensors of E and G, fixed noise maps."""
    import aten_cpu_path as A
    from swapping_autoencoder_pytorch_amd.networks.encoder import StyleGAN2ResnetEncoder
    from swapping_autoencoder_pytorch_amd.networks.generator import StyleGAN2ResnetGenerator
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.stylegan2_layers import NoiseInjection
    opt = make_options("church256", batch_size=16, num_gpus=1)
    b = 16
    ours = _Reconstruction(StyleGAN2ResnetEncoder(opt), StyleGAN2ResnetGenerator(opt)).to(DEV)
    ref32 = _Reconstruction(A.EncoderCPU(opt), A.GeneratorCPU(opt)).to(DEV)
    sp, dp = list(ref32.parameters()), list(ours.parameters())
    assert [tuple(p.shape) for p in sp] == [tuple(p.shape) for p in dp]
    g = torch.Generator().manual_seed(13)
    with torch.no_grad():                 # the reference's own initialisation (N(0, 1) weights), biases / noise strengths made non-zero
        for a_, b_ in zip(sp, dp):
            v = torch.randn(a_.shape, generator=g) * (1.0 if a_.dim() > 1 and tuple(a_.shape) != (1, 3, 1, 1) else 0.2)
            a_.copy_(v)
            b_.copy_(v)
    ref64 = _Reconstruction(A.EncoderCPU(opt), A.GeneratorCPU(opt)).to(DEV).double()
    ref64.load_state_dict({k: v.double() for k, v in ref32.state_dict().items()})
    x = (torch.rand(b, 3, 256, 256, generator=g) * 2 - 1).to(DEV)
    with torch.no_grad():
        ours(x[:1])                       # records every NoiseInjection's map size
    mine_noise = [m for m in ours.modules() if isinstance(m, NoiseInjection)]
    ref_noise = [[m for m in r.modules() if isinstance(m, A.StyledConvCPU)] for r in (ref32, ref64)]
    assert len(mine_noise) == len(ref_noise[0]) == len(ref_noise[1]) > 0
    for i, m in enumerate(mine_noise):
        z = torch.randn(b, 1, m.image_size[2], m.image_size[3], generator=g).to(DEV)
        m.fixed_noise = z
        ref_noise[0][i].fixed_noise = z
        ref_noise[1][i].fixed_noise = z
    t = torch.randn(b, 3, 256, 256, generator=g).to(DEV)
    err_o, err_a = _grads_and_errors(ref64, ref32, ours, [x], lambda m, i: m(i[0]), lambda m, i: m(i[0]), t)
    _check("Encoder -> Generator reconstruction 16x3x256x256 vs ATen restatement in double (cuda:0)", conv_math, err_o, err_a)
