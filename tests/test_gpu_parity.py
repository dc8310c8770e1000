"""GPU suite: the real gfx950 library behind our host-side mirror vs the reference's golden
outputs (ops, layers incl. second-order gradients, the four networks, and four optimiser steps of
the alternating D/G driver).  Tolerances are fp32 relative-to-max per tensor."""
import pytest
import torch

import parity_common as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True, params=["f32", "bf16x6"])
def _native_lib_loaded(request):
    """Every check of this module runs under both conv arithmetics (include/sae_hip.h:
    sae_set_conv_math); the golden tolerances are the same for both."""
    assert torch.cuda.is_available()
    from swapping_autoencoder_pytorch_amd import hip_lib
    lib = hip_lib.get()
    assert lib.prefix == "sae_" and lib.device_only and lib.path.endswith("libsae_hip.so")
    hip_lib.set_conv_math(request.param)
    assert hip_lib.get_conv_math() == request.param
    yield
    hip_lib.set_conv_math("f32")


def test_ops():
    P.check_ops(DEV)


@pytest.mark.parametrize("name", sorted(P.layer_specs()))
def test_layers(name):
    # 1e-4 relative per tensor (north-star tolerance); scalar parameters such as noise.weight are
    # long cancelling fp32 sums
    P.check_layer(name, DEV, tol=1e-4)


def test_micro_networks_forward():
    P.check_micro_forward(DEV)


def test_micro_training_steps(_native_lib_loaded):
    # measured on the MI355X (profiles/r2_step_parity.json): with the exact-fp32 kernels the largest relative loss
    # deviation over the four steps is ~1e-7 and the largest gradient-norm deviation ~3e-6, the same as the CPU back
    # ends and as the oracle under 1e-6 input noise -> held to 5e-6 / 2e-5 (the 5e-3 allowance of round 1 is gone).
    from swapping_autoencoder_pytorch_amd import hip_lib
    if hip_lib.get_conv_math() == "f32":
        P.check_micro_steps(DEV, loss_tol=5e-6, grad_tol=2e-5)
    else:
        # bf16x6: same class (9e-7 / 4e-6 measured) EXCEPT when its different rounding moves a leaky-ReLU input across
        # zero.  In the reference's run of this recipe one unit of D's head (final_linear.0, 10 x 512 units) sits
        # 5e-8 from zero; with bf16x6 (or any other 1e-7-level re-association upstream) it takes the other branch, its
        # derivative changes from sqrt(2) to 0.2 sqrt(2), and D's gradient norms move by up to 1.5e-3 at step 0 (one unit
        # of 5120 ~ 1/sqrt(5120)); Adam(beta1 = 0) then separates the trajectories at the 1e-4 level in the losses.
        # Every recipe has such units (10 M leaky-ReLU inputs per step: the smallest |x| / rms is ~1e-8), so the
        # checks state the size of that one-unit effect instead of pretending it cannot happen: the forward of step 0
        # (losses) stays at 5e-6, its gradient norms at the one-unit level, and the steps after the first Adam update
        # — which turns the changed gradient signs of D's small entries into full +-lr moves — are only required to
        # stay on the same trajectory to 2e-3 in the losses (measured 2.2e-4 / 2.0e-4 / 6.5e-4, profiles/r2_step_parity.json)
        m = P.measure_micro_steps(DEV)
        assert m["steps"][0]["max_loss_dev"] <= 5e-6, m["steps"][0]["loss_dev"]
        assert m["steps"][0]["max_grad_norm_dev"] <= 5e-3, (m["steps"][0]["worst_grad"], m["steps"][0]["max_grad_norm_dev"])
        for row in m["steps"][1:]:
            assert row["max_loss_dev"] <= 2e-3, (row["step"], row["loss_dev"])
        assert m["max_param_norm_dev_after"] <= 5e-2


def test_cpu_tensor_is_refused():
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import upfirdn2d
    with pytest.raises(hip_lib.SaeError):
        upfirdn2d(torch.zeros(1, 2, 4, 4), torch.ones(4, 4) / 16, pad=(2, 2))
