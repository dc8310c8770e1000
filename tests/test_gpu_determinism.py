"""GPU: the train step is bitwise reproducible (no atomics anywhere: fixed-order reductions in the kernels, gather
formulations for the reflection-pad and random-crop backward).  Two runs from the same seed must produce
IDENTICAL loss values, under both conv arithmetics."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mode, iters=3):
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer
    hip_lib.set_conv_math(mode)
    try:
        opt = make_options("tiny32", batch_size=4, num_gpus=1)
        torch.manual_seed(0)
        model = create_model(opt)
        optimizer = create_optimizer(opt, model)
        torch.manual_seed(1)
        g = torch.Generator(device="cuda").manual_seed(2)
        out = []
        for i in range(iters):
            for _ in range(2):
                x = torch.rand(4, 3, opt.crop_size, opt.crop_size, device="cuda", generator=g) * 2 - 1
                losses = optimizer.train_one_step({"real_A": x}, i)
                out.append({k: float(v) for k, v in losses.items()})
        return out
    finally:
        hip_lib.set_conv_math("f32")


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
def test_two_runs_are_bit_identical(mode):
    a, b = _run(mode), _run(mode)
    assert a == b
    assert all(v == v for call in a for v in call.values())      # no NaN
