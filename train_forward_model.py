#!/usr/bin/env python3
"""Drop-in entry point with the reference's name and CLI (`python train_forward_model.py --config-file
config/default.yaml ...`); the implementation is ndivplanning_amd/train_forward_model.py."""
import models.forward_encoder  # noqa: F401  (binds the reference class path for the checkpoints)
from ndivplanning_amd.train_forward_model import denorm, main, norm, train  # noqa: F401

if __name__ == "__main__":
    main()
