"""ndivplanning_amd -- MI355X-native GAN-training hot path of goodmattg/ndivplanning.

Host-side mirror of the reference's Python surface for this path
(`models.gan`, `diversity`, `train_gan`) on top of libndp_hip.so, a C-ABI
library of hand-written gfx950 kernels (include/ndp.h).  There is no CPU
fallback: every compute entry point raises if the HIP library or a GPU is
missing.
"""
__version__ = "0.1.0"

import os as _os

# HSA reads this once, when the runtime initialises (the first torch.cuda / HIP call of the process): the host
# driver of this platform supports dmabuf IPC only, and both the hipIpc peer-to-peer gradient exchange and RCCL
# fail with `hipIpcGetMemHandle: invalid argument` without it.  Importing the package before touching the GPU
# is therefore enough; a launcher that initialises HIP first must export the variable itself.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
