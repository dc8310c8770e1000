"""The two random draws of the training step, in one place: uniform numbers for the random
crops (util/util.py:326,335-336) and the per-layer N(0,1) noise maps of the generator
(stylegan2_layers.py:340-342).  Parity tests replace these two functions to feed the CPU-generated
stream of the oracle run to a GPU model."""
import torch


def rand(shape, device):
    return torch.rand(*shape, device=device)


def randn_like_image(image, channels=1):
    b, _, h, w = image.shape
    return image.new_empty(b, channels, h, w).normal_()
