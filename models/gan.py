"""Reference-name shim: `from models.gan import Decoder, Discriminator`.

Importing through this name also makes whole-module checkpoints interchangeable with the
reference's: torch.save(decoder) then records the class as `models.gan.Decoder`
(train_gan.py:254-266), which is the path the reference's evaluation scripts unpickle
(control_evaluation.py:175-176)."""
from ndivplanning_amd.models.gan import (Decoder, Discriminator, collapse_batch, normal_init,  # noqa: F401
                                         uncollapse_batch)

Decoder.__module__ = __name__
Discriminator.__module__ = __name__
