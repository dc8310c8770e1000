"""GPU: the train step is bitwise reproducible (no atomics anywhere: fixed-order reductions in the kernels, gather
formulations for the reflection-pad and random-crop backward).  Two runs from the same seed must produce
IDENTICAL loss values, under both conv arithmetics."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mode, iters=3, two_streams=None, r1_every=None):
    import os
    from swapping_autoencoder_pytorch_amd import hip_lib
    if two_streams is not None:
        os.environ["SAE_TWO_STREAMS"] = "1" if two_streams else "0"
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer
    hip_lib.set_conv_math(mode)
    try:
        opt = make_options("tiny32", batch_size=4, num_gpus=1)
        if r1_every:
            opt.R1_once_every = r1_every
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
        out.append({k: float(v.double().sum()) for k, v in model.singlegpu_model.state_dict().items() if v.dtype.is_floating_point})
        return out
    finally:
        hip_lib.set_conv_math("f32")
        os.environ.pop("SAE_TWO_STREAMS", None)


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
def test_two_runs_are_bit_identical(mode):
    a, b = _run(mode), _run(mode)
    assert a == b
    assert all(v == v for call in a for v in call.values())      # no NaN


def test_two_streams_change_no_value():
    """swapping_autoencoder_pytorch_amd/streams.py: the generator's two passes and the two discriminators on two HIP streams
    (forward, and through autograd backward) give the SAME bits as everything on one stream -- losses of every call incl. a
    lazy-R1 call, and every parameter after the updates."""
    one, two = _run("f32", two_streams=False, r1_every=2), _run("f32", two_streams=True, r1_every=2)
    assert any("D_R1" in call for call in one)
    assert one == two
