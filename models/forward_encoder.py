"""Reference-name shim: `from models.forward_encoder import ForwardAutoencoder` (the class path inside the
reference's forward_autoencoder_*.pt pickles, train_forward_model.py:157-163; control_evaluation.py:177-180)."""
from ndivplanning_amd.models.forward_encoder import Decoder, Encoder, ForwardAutoencoder, normal_init  # noqa: F401

ForwardAutoencoder.__module__ = __name__
Encoder.__module__ = __name__
Decoder.__module__ = __name__
