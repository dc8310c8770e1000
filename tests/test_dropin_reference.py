"""The drop-in boundary, exercised with the reference's OWN code on top: with our operator and
layer modules pre-seeded under the reference's import names (dropin.preseed), the reference's
networks / model / optimizer (imported unmodified from /root/reference) must reproduce the golden
numbers they produced on their own layers.  Needs the reference checkout, which exists only in
the build container (never on the GPU box) -> skipped elsewhere."""
import os
import sys

import pytest
import torch

import parity_common as P
import ref_shims
from param_recipe import MICRO, fill_params, uniform_images

pytestmark = pytest.mark.skipif(not os.path.isdir(ref_shims.REF), reason="reference checkout not present")


def _ref_options(**over):
    from options import TrainOptions
    argv = sys.argv
    sys.argv = ["train.py", "--name", "dropin", "--dataset_mode", "imagefolder"]
    try:
        opt = TrainOptions().gather_options()
    finally:
        sys.argv = argv
    opt.isTrain = True
    for k, v in over.items():
        setattr(opt, k, v)
    return opt


def test_reference_training_loop_on_our_layers(oracle_lib):
    already = [m for m in sys.modules if m == "models" or m.startswith("models.")]
    assert not already, "reference modules imported before pre-seeding: %s" % already[:3]
    ref_shims.install()
    from swapping_autoencoder_pytorch_amd import dropin
    dropin.preseed()
    import models                       # the reference's package
    import optimizers
    import models.networks              # parent package from the reference checkout
    layers = sys.modules["models.networks.stylegan2_layers"]
    import swapping_autoencoder_pytorch_amd.stylegan2_layers as ours
    assert layers is ours               # the reference now builds its networks from our layers
    _, info = P.golden()
    opt = _ref_options(**MICRO)
    with P.backend(oracle_lib):
        torch.manual_seed(0)
        model = models.create_model(opt)
        net = model.singlegpu_model
        assert type(net.E.FromRGB).__module__ == ours.__name__
        fill_params(net, seed=3)
        optimizer = optimizers.create_optimizer(opt, model)
        for it in range(2):              # one D call, one G call of the reference's driver
            torch.manual_seed(1000 + it)
            losses = optimizer.train_one_step({"real_A": uniform_images(4, 32, 600 + it)}, it)
            want = info["micro_steps"]["step%d" % it]
            assert set(losses) == set(want)
            for k, v in want.items():
                assert abs(float(losses[k]) - v) <= 2e-4 * max(1.0, abs(v)), (it, k, float(losses[k]), v)
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.") or m == "optimizers"
                 or m.startswith("optimizers.")]:
        del sys.modules[name]           # do not leak the reference's packages into other tests
