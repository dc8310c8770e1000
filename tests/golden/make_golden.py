"""Generate the golden fixtures of tests/golden/ by running the REFERENCE's own Python modules
(/root/reference, read-only) on CPU through its documented native-fallback path
(upfirdn2d.py:162-222, fused_act.py:93-96).  Run in the build container only:

    python tests/golden/make_golden.py

The reference cannot import on ROCm/CPU-only PyTorch as shipped (util.is_custom_kernel_supported
parses torch.version.cuda, util/util.py:432-436) and imports packages absent here at module
top level; the three non-invasive shims of SURVEY.md §7.1 are applied before import.  Nothing of
the reference is copied: only its OUTPUTS on seeded inputs are stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
REF = "/root/reference"
sys.path.insert(0, TESTS)
sys.path.insert(0, ROOT)

from param_recipe import MICRO, fill_params, seeded, uniform_images  # noqa: E402


class _Stub(types.ModuleType):
    """import-only stand-in: any attribute is a dummy class (never executed on the hot path)"""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return type(item, (), {"__init__": lambda self, *a, **k: None})


def import_reference():
    torch.version.cuda = "0.0"   # the gate returns False -> native fallback, instead of raising
    for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.models",
                 "torchvision.datasets", "dominate", "dominate.tags", "func_timeout", "visdom", "GPUtil", "cv2", "lmdb"]:
        if name not in sys.modules:
            m = _Stub(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["visdom"].Visdom = type("Visdom", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["func_timeout"].func_timeout = lambda *a, **k: None
    sys.modules["func_timeout"].FunctionTimedOut = Exception
    sys.path.insert(0, REF)


def ref_options(**overrides):
    """The reference's own parser defaults (options/__init__.py:55-97) + overrides."""
    from options import TrainOptions
    argv = sys.argv
    sys.argv = ["train.py", "--name", "golden", "--dataset_mode", "imagefolder"]
    try:
        opt = TrainOptions().gather_options()
    finally:
        sys.argv = argv
    opt.isTrain = True
    for k, v in overrides.items():
        assert hasattr(opt, k), k
        setattr(opt, k, v)
    return opt


def t2n(t):
    return t.detach().cpu().numpy().astype(np.float32)


def golden_ops(out):
    from models.networks.stylegan2_op.upfirdn2d import upfirdn2d_native
    from models.networks.stylegan2_op.fused_act import fused_leaky_relu
    cases = [
        ((2, 3, 16, 16), [1, 3, 3, 1], 1, 1, (2, 2)),
        ((2, 3, 16, 16), [1, 3, 3, 1], 1, 1, (1, 1)),
        ((1, 4, 19, 19), [1, 2, 1], 1, 1, (0, 0)),
        ((1, 4, 12, 12), [1, 2, 1], 1, 1, (1, 0)),
        ((2, 2, 9, 9), [1], 1, 1, (0, 0)),
        ((2, 2, 8, 8), [1, 3, 3, 1], 2, 1, (2, 1)),      # Upsample
        ((2, 2, 16, 16), [1, 3, 3, 1], 1, 2, (1, 1)),    # Downsample
        ((1, 2, 33, 33), [1, 3, 3, 1], 1, 1, (1, 1)),
    ]
    meta = []
    for i, (shape, k1d, up, down, pad) in enumerate(cases):
        x = seeded(shape, 100 + i).requires_grad_()
        k = torch.tensor(k1d, dtype=torch.float32)
        k = k[None, :] * k[:, None]
        k = k / k.sum() * (up ** 2)
        y = upfirdn2d_native(x, k, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        w = seeded(y.shape, 200 + i)
        gx, = torch.autograd.grad((y * w).sum(), x)
        out["upfirdn_%d_y" % i] = t2n(y)
        out["upfirdn_%d_gx" % i] = t2n(gx)
        meta.append(dict(shape=shape, k=k1d, up=up, down=down, pad=pad))
    out["upfirdn_meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    for i, shape in enumerate([(2, 5, 7, 7), (3, 6)]):
        x = seeded(shape, 300 + i).requires_grad_()
        b = seeded((shape[1],), 310 + i).requires_grad_()
        y = fused_leaky_relu(x, b)
        w = seeded(shape, 320 + i)
        gx, gb = torch.autograd.grad((y * w).sum(), [x, b])
        out["lrelu_%d_y" % i], out["lrelu_%d_gx" % i], out["lrelu_%d_gb" % i] = t2n(y), t2n(gx), t2n(gb)


def golden_layers(out):
    import models.networks.stylegan2_layers as L
    from models.networks.generator import UpsamplingResnetBlock
    specs = {
        "g_upblock": (lambda: UpsamplingResnetBlock(6, 10, 16, use_noise=True), (2, 6, 8, 8), True),
        "modconv": (lambda: L.ModulatedConv2d(6, 10, 3, 16), (2, 6, 8, 8), True),
        "modconv_up": (lambda: L.ModulatedConv2d(6, 10, 3, 16, upsample=True), (2, 6, 8, 8), True),
        "modconv_nodemod_1x1": (lambda: L.ModulatedConv2d(6, 3, 1, 16, demodulate=False), (2, 6, 8, 8), True),
        "styledconv_up": (lambda: L.StyledConv(6, 10, 3, 16, upsample=True), (2, 6, 8, 8), True),
        "torgb": (lambda: L.ToRGB(6, 16), (2, 6, 8, 8), True),
        "convlayer_down": (lambda: L.ConvLayer(6, 10, 3, downsample=True), (2, 6, 16, 16), False),
        "convlayer_refpad": (lambda: L.ConvLayer(6, 10, 3, reflection_pad=True), (2, 6, 9, 9), False),
        "convlayer_nobias": (lambda: L.ConvLayer(6, 10, 1, activate=True, bias=False), (2, 6, 9, 9), False),
        "resblock": (lambda: L.ResBlock(6, 10), (2, 6, 16, 16), False),
        "resblock_e": (lambda: L.ResBlock(6, 10, [1, 2, 1], reflection_pad=True), (2, 6, 16, 16), False),
        "resblock_nodown": (lambda: L.ResBlock(6, 10, downsample=False), (2, 6, 8, 8), False),
        "equallinear_act": (lambda: L.EqualLinear(12, 7, activation="fused_lrelu"), (5, 12), False),
        "equallinear": (lambda: L.EqualLinear(12, 7, bias_init=1, lr_mul=0.5), (5, 12), False),
    }
    for name, (ctor, xshape, styled) in specs.items():
        torch.manual_seed(0)
        m = ctor()
        fill_params(m, seed=7)
        x = seeded(xshape, 400).requires_grad_()
        torch.manual_seed(11)          # noise draws of StyledConv
        if styled:
            s = seeded((xshape[0], 16), 401).requires_grad_()
            y = m(x, s)
        else:
            s = None
            y = m(x)
        w = seeded(y.shape, 402)
        params = [p for p in m.parameters()]
        grads = torch.autograd.grad((y * w).sum(), [x] + ([s] if styled else []) + params, allow_unused=True)
        out["layer_%s_y" % name] = t2n(y)
        out["layer_%s_gx" % name] = t2n(grads[0])
        if styled:
            out["layer_%s_gs" % name] = t2n(grads[1])
        off = 2 if styled else 1
        for (pn, _), g in zip(m.named_parameters(), grads[off:]):
            if g is not None:
                out["layer_%s_gp_%s" % (name, pn)] = t2n(g)
        # second order (R1-style) for the D-side layers
        if not styled and len(xshape) == 4:
            x2 = seeded(xshape, 400).requires_grad_()
            y2 = m(x2)
            g1, = torch.autograd.grad(y2.sum(), x2, create_graph=True)
            pen = g1.pow(2).sum()
            gp = torch.autograd.grad(pen, params, allow_unused=True)
            for (pn, _), g in zip(m.named_parameters(), gp):
                if g is not None:
                    out["layer_%s_r1gp_%s" % (name, pn)] = t2n(g)


def grad_summary(params_named):
    """per-parameter (l2 norm, first 4 entries) – compact but sensitive"""
    res = {}
    for n, p in params_named:
        if p.grad is None:
            continue
        g = p.grad.detach().double().flatten()
        res[n] = [float(g.norm())] + [float(v) for v in g[:4]]
    return res


def golden_model(out, info):
    import models
    import optimizers
    # key/shape inventory of the church preset (meta device: no memory)
    opt = ref_options(crop_size=256, load_size=256, num_gpus=0, patch_use_aggregation=False)
    with torch.device("meta"):
        m = models.find_model_using_name(opt.model)(opt)
        m.initialize()
    info["church_state_dict"] = {k: list(v.shape) for k, v in m.state_dict().items()}

    opt = ref_options(**MICRO)
    torch.manual_seed(0)
    model = models.create_model(opt)
    net = model.singlegpu_model
    fill_params(net, seed=3)
    info["micro_state_dict"] = {k: list(v.shape) for k, v in net.state_dict().items()}
    real = uniform_images(4, 32, 500)

    # network forwards
    torch.manual_seed(21)
    sp, gl = net.E(real)
    out["micro_E_sp"], out["micro_E_gl"] = t2n(sp), t2n(gl)
    torch.manual_seed(22)
    rec = net.G(sp, gl)
    out["micro_G_rec"] = t2n(rec)
    out["micro_D_pred"] = t2n(net.D(real))
    torch.manual_seed(23)
    crops = net.get_random_crops(real)
    out["micro_crops"] = t2n(crops)
    feat = net.Dpatch.extract_features(crops, aggregate=True)
    out["micro_Dpatch_feat"] = t2n(feat)
    out["micro_Dpatch_pred"] = t2n(net.Dpatch.discriminate_features(feat, feat.flip(0)))

    # one D step, one G step, one D step with R1 (R1_once_every = 2), one G step: losses + grads
    optimizer = optimizers.create_optimizer(opt, model)
    steps = {}
    for it in range(4):
        torch.manual_seed(1000 + it)
        data = {"real_A": uniform_images(4, 32, 600 + it)}
        losses = optimizer.train_one_step(data, it)
        steps["step%d" % it] = {k: float(v) for k, v in losses.items()}
        steps["step%d_grads" % it] = grad_summary(net.named_parameters())
    info["micro_steps"] = steps
    # parameter checksum after the 4 optimiser steps
    info["micro_param_norms_after"] = {k: float(v.double().norm()) for k, v in net.state_dict().items()
                                       if v.dtype.is_floating_point}


def main():
    import_reference()
    torch.set_num_threads(8)
    out, info = {}, {}
    golden_ops(out)
    golden_layers(out)
    golden_model(out, info)
    np.savez_compressed(os.path.join(HERE, "reference_outputs.npz"), **out)
    info["torch"] = torch.__version__
    with open(os.path.join(HERE, "reference_info.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "reference_outputs.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
