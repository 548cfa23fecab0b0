#!/usr/bin/env python3
"""Drop-in entry point with the reference's name and CLI (`python train_gan.py --config-file
config/default.yaml ...`); the implementation is ndivplanning_amd/train_gan.py."""
import models.gan  # noqa: F401  (binds the reference class paths `models.gan.*` for the checkpoints)
import models.image_autoencoder  # noqa: F401
from ndivplanning_amd.train_gan import denorm, main, norm, train  # noqa: F401

if __name__ == "__main__":
    main()
