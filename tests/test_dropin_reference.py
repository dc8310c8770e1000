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
    dropin.preseed("layers")            # operators + layer library only: the reference's own networks and model on top
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


@pytest.mark.parametrize("level", ["layers", "networks", "full"])
def test_reference_driver_at_every_preseed_level(level, oracle_lib):
    """dropin.preseed(level) + patch_util(): the reference's unmodified optimizer, models/__init__.py (create_model,
    MultiGPUModelWrapper / DataParallel) and option parser driving (layers) its own networks and model on our layers,
    (networks) its own model on our networks, (full) our model class under its BaseModel -- four optimiser calls
    (D, G, D + lazy R1, G) against the golden loss dictionaries of the reference's own run.  One subprocess per level
    (tests/dropin_ref_worker.py::preseed_level) so that nothing pre-seeded leaks into this process."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "dropin_ref_worker.py"), "preseed_level", level],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rep = json.loads([l for l in out.stdout.splitlines() if l.startswith("LEVEL-REPORT ")][-1][len("LEVEL-REPORT "):])
    assert rep["level"] == level and rep["worst_loss_dev"] <= 2e-4, rep
    ours = "swapping_autoencoder_pytorch_amd"
    assert rep["layer_module"].startswith(ours)
    assert rep["encoder_module"].startswith(ours) == (level != "layers"), rep
    assert rep["model_is_ours"] == (level == "full") and rep["model_is_basemodel"], rep
    assert rep["normalize_module"].startswith(ours) and rep["crop_module"].startswith(ours), rep
    assert rep["r1_wrapped"] == (level != "full"), rep
