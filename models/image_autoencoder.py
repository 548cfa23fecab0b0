"""Reference-name shim: `from models.image_autoencoder import Encoder` (the class path inside
the reference's encoder.pt pickles, train_gan.py:75)."""
from ndivplanning_amd.models.image_autoencoder import Encoder, normal_init  # noqa: F401

Encoder.__module__ = __name__
