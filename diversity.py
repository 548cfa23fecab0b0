"""Reference-name shim: `import diversity` resolves to the HIP-backed implementation."""
from ndivplanning_amd.diversity import (compute_pair_distance, compute_pair_unnormal_distance,  # noqa: F401
                                        compute_pairwise, compute_pairwise_divergence)
