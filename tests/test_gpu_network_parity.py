"""Network-level parity at BASELINE size on the GPU (church256, B = 16): the image discriminator (16 x 3 x 256 x 256 ->
logits, stylegan2_layers.py:696-763) and one upsampling block of the generator (16 x 256 x 128 x 128 -> 16 x 128 x 256 x
256 with explicit noise maps, generator.py:39-53) of THIS package against the ATen restatement of the reference's code path
(oracle/aten_cpu_path.py, pinned to the reference's own modules in tests/test_dropin_train.py::
test_aten_cpu_path_matches_reference_discriminator) moved to cuda:0 and run in DOUBLE precision — same weights, same input:
the outputs within 1e-4 of the tensor's largest magnitude, the north-star tolerance, under both conv arithmetics; every gradient
tensor within 1e-4 (relative L2) of the MASK-FROZEN double run -- the double restatement evaluated under the leaky-ReLU sign
pattern this package's run took (tests/mask_frozen.py), which removes the one discontinuity of the network and with it the
dependence of the comparison on the box and on MIOpen's solver choice (round 4's red test).  The free double run and the stock
ATen / MIOpen fp32 control are logged per tensor next to it.  The kernel-level full-size cases (test_gpu_fullsize_oracle.py) check each conv class against the
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
    """Per tensor (output, input gradients, every parameter gradient), (max-norm, L2, fraction) errors of
      "ours"            this package vs the double run under THIS package's leaky-ReLU sign patterns (mask_frozen.py): asserted
      "control"         stock ATen / MIOpen fp32 vs the double run under ITS OWN sign patterns: logged
      "ours_free" / "control_free"   both against the free double run (what rounds 3-4 compared): logged
    plus the bookkeeping of the frozen references (activation calls matched, flips imposed)."""
    import aten_cpu_path as A
    from mask_frozen import SavedActivations, frozen_reference_grads
    r64 = [t.detach().double().requires_grad_(True) for t in inputs]
    r32 = [t.detach().clone().requires_grad_(True) for t in inputs]
    mine = [t.detach().clone().requires_grad_(True) for t in inputs]
    y64 = run_ref(ref64, r64)
    with A.ActivationMasks.record() as rec:
        y32 = run_ref(ref32, r32)
    with SavedActivations() as saved:
        yo = run_ours(ours, mine)
    g64 = torch.autograd.grad((y64 * loss_weight.double()).sum(), r64 + list(ref64.parameters()))
    g32 = torch.autograd.grad((y32 * loss_weight).sum(), r32 + list(ref32.parameters()))
    go = torch.autograd.grad((yo * loss_weight).sum(), mine + list(ours.parameters()))
    y64o, g64o, flips_o = frozen_reference_grads(A, ref64, inputs, run_ref, loss_weight, saved=saved)
    y64c, g64c, flips_c = frozen_reference_grads(A, ref64, inputs, run_ref, loss_weight, masks=rec.masks)
    names = ["grad_input%d" % i for i in range(len(inputs))] + ["grad " + n for n, _ in ours.named_parameters()]
    cols = {"ours": (go, g64o), "control": (g32, g64c), "ours_free": (go, g64), "control_free": (g32, g64)}
    errs = {"ours": {"output": _stats(yo, y64o)}, "control": {"output": _stats(y32, y64c)},
            "ours_free": {"output": _stats(yo, y64)}, "control_free": {"output": _stats(y32, y64)}}
    for col, (got, want) in cols.items():
        pooled = {}
        for n, a, c in zip(names, got, want):
            if c.numel() == 1:
                # one-element gradients (the NoiseInjection strengths: ONE cancelling sum over N * C * H * W terms each)
                # are compared as a vector per parameter kind
                kind = "grad *." + ".".join(n.split(".")[-2:]) + " (pooled one-element gradients)"
                pooled.setdefault(kind, ([], []))
                pooled[kind][0].append(a.detach().reshape(1).double())
                pooled[kind][1].append(c.detach().reshape(1).double())
                continue
            errs[col][n] = _stats(a, c)
        for kind, (la, lc) in pooled.items():
            errs[col][kind] = _stats(torch.cat(la), torch.cat(lc))
    book = {"activation calls": len(flips_o), "activation elements": int(saved.elements),
            "matched to a saved activation of this package": saved.matched,
            "unmatched": saved.unmatched, "elements flipped vs the double run (ours)": int(sum(f for f in flips_o if f > 0)),
            "elements flipped vs the double run (control)": int(sum(flips_c))}
    return errs, book


# relative L2 of every gradient tensor against the double run under this package's own sign patterns: the north-star tolerance,
# for both arithmetics (measured r5, profiles/r5_network_parity_tensors.jsonl: f32 worst tensor 2.4e-6; the opt-in bf16x6 arithmetic
# -- six bf16 products per fp32 product, DESIGN.md section 4 (bf16x6) -- 4.8e-5; with the Winograd route on 2.2e-6)
GRAD_TOL = {"f32": 1e-4, "bf16x6": 1e-4}


def _check(case, conv_math, errs, book):
    """The network OUTPUT is continuous in its inputs: within 1e-4 (max norm) of the double run.  GRADIENTS are compared with
    the MASK-FROZEN double run (tests/mask_frozen.py): the double restatement evaluated under the leaky-ReLU sign pattern this
    package's run actually took, which makes every gradient a continuous function of inputs and weights and the north-star
    tolerance assertable directly, on any box.  The free double run and the stock ATen / MIOpen fp32 control (rounds 3-4's
    criterion, whose limit moved with MIOpen's solver choice per box) ride along as logged columns, tensor by tensor
    (gpurun_out/network_parity_tensors.jsonl)."""
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "network_parity_tensors.jsonl"), "a") as f:
            for k in errs["ours"]:
                f.write(json.dumps({"case": case.split(" vs ")[0], "conv_math": conv_math, "tensor": k,
                                    **{col: errs[col][k][:2] for col in errs}}) + "\n")
    worst = max((k for k in errs["ours"] if k != "output"), key=lambda k: errs["ours"][k][1])
    _log({"case": case, "conv_math": conv_math, "tensors": len(errs["ours"]), "output_max_err": errs["ours"]["output"][0],
          **{"max over gradient tensors (max norm, l2): " + col: [max(v[i] for k, v in e.items() if k != "output") for i in range(2)]
             for col, e in errs.items()},
          "worst tensor (l2, frozen)": worst, "its l2 error": errs["ours"][worst][1],
          "control's l2 error on it (frozen to its own masks)": errs["control"][worst][1], **book})
    assert errs["ours"]["output"][0] < TOL and errs["ours_free"]["output"][0] < TOL, errs["ours"]["output"]
    assert book["unmatched"] == [] and book["matched to a saved activation of this package"] == book["activation calls"], book
    bad = {k: v[:2] for k, v in errs["ours"].items() if k != "output" and not v[1] <= GRAD_TOL[conv_math]}
    assert not bad, bad
    # ... and ONE assertion on the FREE run, so that a real sign bug cannot hide behind the frozen masks: this package's forward
    # pass may put no more leaky-ReLU inputs on the other side of zero than the double run does than twice what the stock ATen /
    # MIOpen fp32 run does (+ a floor of one element per million for networks where the control happens to flip none).  A wrong
    # sign, bias or scale anywhere upstream of an activation flips a sizeable FRACTION of its elements, not a handful.
    flips, control = book["elements flipped vs the double run (ours)"], book["elements flipped vs the double run (control)"]
    assert flips <= 2 * control + max(8, book["activation elements"] // 1000000), (flips, control, book["activation elements"])


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
    # per-image loss weights of ONE sign: with weights that sum to zero (rounds 3-4: linspace(-1, 1)) the last layer's weight
    # gradient sum_n w_n h_n cancels ~30x on images that are all uniform noise, and the forward rounding of h (3e-6) alone puts
    # that one tensor at 1.04e-4 for this package and 1.07e-4 for stock ATen alike (profiles/r5_network_parity_tensors.jsonl)
    w = torch.linspace(0.25, 1.0, 16, device=DEV).view(16, 1)
    errs, book = _grads_and_errors(ref64, ref32, ours, [x], lambda m, i: m(i[0]), lambda m, i: m(i[0]), w)
    _check("Discriminator 16x3x256x256 vs ATen restatement in double (cuda:0)", conv_math, errs, book)


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

    errs, book = _grads_and_errors(ref64, ref32, ours, [x, style], run_ref, lambda m, i: m(i[0], i[1]), t)
    _check("UpsamplingResnetBlock 16x256x128x128 -> 16x128x256x256 vs ATen restatement in double (cuda:0)", conv_math, errs, book)


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
    errs, book = _grads_and_errors(ref64, ref32, ours, [x], lambda m, i: m(i[0]), lambda m, i: m(i[0]), t)
    _check("Encoder -> Generator reconstruction 16x3x256x256 vs ATen restatement in double (cuda:0)", conv_math, errs, book)


class _PatchPair(torch.nn.Module):
    """(reference patches, target patches) -> Dpatch logits: extract_features on both (the reference side aggregated over the
    crops of an image), then discriminate_features -- swapping_autoencoder_model.py:95-114, patch_discriminator.py:146-171"""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, ref_patches, target_patches):
        f_ref = self.net.extract_features(ref_patches, aggregate=True)
        f_tgt = self.net.extract_features(target_patches)
        return self.net.discriminate_features(f_ref, f_tgt)


def test_patch_discriminator_church256_b16_vs_aten_restatement(conv_math):
    """The whole patch discriminator at the BASELINE widths and patch size: 8 images x 8 crops of 128 x 128 per side (half the
    batch of a discriminator call: the double restatement of 2 x 128 patches alone took four minutes per arithmetic), aggregation
    over the reference crops, the pair MLP -- outputs, both patch gradients (what the generator step and the patch R1 penalty
    consume) and every parameter gradient."""
    import aten_cpu_path as A
    from swapping_autoencoder_pytorch_amd.networks.patch_discriminator import StyleGAN2PatchDiscriminator
    from swapping_autoencoder_pytorch_amd.options import make_options
    opt = make_options("church256", batch_size=16, num_gpus=1)
    b, t, ps = 8, opt.patch_num_crops, opt.patch_size
    ours = _PatchPair(StyleGAN2PatchDiscriminator(opt)).to(DEV)
    ref32 = _PatchPair(A.PatchDiscriminatorCPU(opt)).to(DEV)
    _copy_params(ref32, ours, 17)
    ref64 = _PatchPair(A.PatchDiscriminatorCPU(opt)).to(DEV).double()
    ref64.load_state_dict({k: v.double() for k, v in ref32.state_dict().items()})
    g = torch.Generator().manual_seed(19)
    pr = (torch.rand(b, t, 3, ps, ps, generator=g) * 2 - 1).to(DEV)
    pt = (torch.rand(b, t, 3, ps, ps, generator=g) * 2 - 1).to(DEV)
    w = torch.linspace(0.25, 1.0, b * t, device=DEV).view(b * t, 1)       # one sign: see the discriminator test
    run = lambda m, i: m(i[0], i[1])
    errs, book = _grads_and_errors(ref64, ref32, ours, [pr, pt], run, run, w)
    _check("PatchDiscriminator 2 x (8x8)x3x128x128 vs ATen restatement in double (cuda:0)", conv_math, errs, book)
