"""E / G / D / Dpatch of the Swapping Autoencoder on the MI355X layer library.

Same class names, option flags, module/parameter names (checkpoint keys) and forward semantics as
the reference's models/networks/{encoder,generator,discriminator,patch_discriminator}.py."""
from .base_network import BaseNetwork
from .encoder import StyleGAN2ResnetEncoder
from .generator import StyleGAN2ResnetGenerator
from .discriminator import StyleGAN2Discriminator
from .patch_discriminator import StyleGAN2PatchDiscriminator

_REGISTRY = {
    ("StyleGAN2Resnet", "encoder"): StyleGAN2ResnetEncoder,
    ("StyleGAN2Resnet", "generator"): StyleGAN2ResnetGenerator,
    ("StyleGAN2", "discriminator"): StyleGAN2Discriminator,
    ("StyleGAN2", "patch_discriminator"): StyleGAN2PatchDiscriminator,
}


def find_network_using_name(network_name, mode):
    try:
        return _REGISTRY[(network_name, mode)]
    except KeyError:
        raise ValueError("no %s network named %r (available: %s)" % (mode, network_name, sorted(_REGISTRY)))


def create_network(opt, network_name, mode, verbose=False):
    """models/networks/__init__.py:39-46"""
    if network_name is None:
        return None
    net = find_network_using_name(network_name, mode)(opt)
    if verbose:
        net.print_architecture(verbose=True)
    return net
