"""ndivplanning_amd -- MI355X-native GAN-training hot path of goodmattg/ndivplanning.

Host-side mirror of the reference's Python surface for this path
(`models.gan`, `diversity`, `train_gan`) on top of libndp_hip.so, a C-ABI
library of hand-written gfx950 kernels (include/ndp.h).  There is no CPU
fallback: every compute entry point raises if the HIP library or a GPU is
missing.
"""
__version__ = "0.1.0"
