"""Data parallelism over the batch dimension.

The GAN step shards naturally over trajectories: G and D act row-wise, NDiv pairs exist
only among the K samples of one row, so rank r trains rows [r*FLAT/W, (r+1)*FLAT/W) and the
only exchange is the SUM of the D gradient (58,305 f32) and of the G gradient (83,780 f32),
one flat all-reduce each per step, over RCCL (torch.distributed backend "nccl" on ROCm).

Scaling rule that makes W ranks reproduce the single-process global batch:
  * BCE terms are MEANS over M rows (train_gan.py:174-190): each rank divides by the GLOBAL
    row count (`inv_m_global`), so the summed gradients are the global-mean gradients;
  * the NDiv term is a SUM over rows (diversity.py:41): no scaling, the sum of the ranks'
    gradients is the global gradient (averaging would shrink pairwise_div_factor by 1/W);
  * every rank applies the same Adam update to its replica; loss shares add up.

`run_step` is the backend-agnostic order of one iteration (train_gan.py:172-203); the HIP
trainer and the oracle-backed test double implement the four phase methods.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


class _stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator is created; anything
    that parses this process' stdout (bench.py's one JSON line) must not see it."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import sys
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def init_process_group(device=None, force=False):
    """Initialise torch.distributed for one process per GPU (RCCL) or, without a GPU, gloo.
    `force` creates a one-rank group too (to exercise the data-parallel path on one GPU)."""
    rank, world, _ = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this platform
        backend = os.environ.get("NDP_DIST_BACKEND")      # tests force gloo to run 2 ranks on one GPU
        if backend is None:
            backend = "nccl" if device is not None and torch.device(device).type == "cuda" else "gloo"
        kw = {}
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            kw = dict(rank=0, world_size=1)
        if backend == "nccl":
            with _stdout_to_stderr():
                dist.init_process_group(backend="nccl", device_id=torch.device(device), **kw)
                warm = torch.zeros(1, device=torch.device(device))
                dist.all_reduce(warm)                       # creates the communicator (and the banner) now
                torch.cuda.synchronize(torch.device(device))
        else:
            dist.init_process_group(backend=backend, **kw)
    return rank, world


def shard_bounds(n_global, rank, world):
    """Contiguous, equal shards: rows [lo, hi) of rank `rank`.  n_global must divide evenly
    (the loss-scaling rule assumes equal shard sizes only through inv_m_global, which it does
    not -- but equal shards keep the step time balanced); a remainder is rejected loudly."""
    if n_global % world != 0:
        raise ValueError("global batch of %d rows does not split evenly over %d ranks" % (n_global, world))
    per = n_global // world
    return rank * per, (rank + 1) * per


def sum_all_reduce(group=None):
    """reduce_fn for the trainer: in-place SUM all-reduce of a flat gradient."""
    def reduce_fn(flat_grad):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return reduce_fn


def run_step(backend, reduce_fn, discrim_steps=1):
    """One training iteration in data-parallel order.  `backend` provides
    d_grads(first) -> flat D gradient, apply_d(grad), g_grads() -> flat G gradient, apply_g(grad)."""
    for it in range(discrim_steps):
        grad = backend.d_grads(it == 0)
        reduce_fn(grad)
        backend.apply_d(grad)
    grad = backend.g_grads()
    reduce_fn(grad)
    backend.apply_g(grad)


def reduce_loss_shares(shares, device=None):
    """Sum per-rank loss shares (each rank holds its part of the global D/G mean and of the
    NDiv sum) into the global values, on every rank."""
    t = torch.tensor(list(shares), dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
