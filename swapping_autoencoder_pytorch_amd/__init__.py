"""MI355X-native hot path of the Swapping-Autoencoder GAN training step."""
__all__ = ["hip_lib"]
