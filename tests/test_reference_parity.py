"""CPU suite: our host-side mirror vs the reference's golden outputs, with the CPU oracle (pins
the oracle to the reference) and with the emulated HIP kernels behind the C-ABI."""
import pytest
import torch

import parity_common as P


def test_oracle_ops_match_reference(oracle_lib):
    with P.backend(oracle_lib):
        P.check_ops("cpu")


def test_emulated_kernels_ops_match_reference(emu_lib):
    with P.backend(emu_lib):
        P.check_ops("cpu")


@pytest.mark.parametrize("name", sorted(P.layer_specs()))
def test_layers_oracle(oracle_lib, name):
    with P.backend(oracle_lib):
        P.check_layer(name, "cpu")


@pytest.mark.parametrize("name", sorted(P.layer_specs()))
def test_layers_emulated_kernels(emu_lib, name):
    with P.backend(emu_lib):
        P.check_layer(name, "cpu")


def test_checkpoint_key_inventory():
    P.check_state_dict_inventory()


def test_micro_networks_forward(oracle_lib):
    with P.backend(oracle_lib):
        P.check_micro_forward("cpu")


def test_micro_training_steps(oracle_lib):
    with P.backend(oracle_lib):
        P.check_micro_steps("cpu")


def test_product_has_no_cpu_path():
    """Without a test back end injected, CPU tensors must be refused loudly (no silent fallback)."""
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.stylegan2_op import fused_leaky_relu, upfirdn2d
    x = torch.zeros(1, 2, 4, 4)
    with pytest.raises(hip_lib.SaeError):
        upfirdn2d(x, torch.ones(4, 4) / 16, pad=(2, 2))
    with pytest.raises(hip_lib.SaeError):
        fused_leaky_relu(x, torch.zeros(2))
