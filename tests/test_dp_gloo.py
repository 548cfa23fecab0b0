"""world_size-2 data-parallel test on CPU (gloo): the DP driver (ndivplanning_amd/dp.py) with
the loss-scaling rule of SURVEY.md section 8e reproduces the single-process global-batch step.
The compute backend here is the oracle (tests may use it); the HIP trainer implements the same
four-phase protocol and is checked on the GPU box."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat(params):
    return torch.cat([p.reshape(-1) for p in params.values()])


def _unflat(flat, like):
    out, off = {}, 0
    for k, v in like.items():
        out[k] = flat[off:off + v.numel()].view_as(v)
        off += v.numel()
    return out


class OracleBackend:
    """dp.run_step backend on top of oracle.StepMath for one rank's shard."""

    def __init__(self, O, g, d, codes, actions, noise, inv_m_global):
        self.sm = O.StepMath(g, d)
        self.codes, self.actions, self.noise, self.inv_m = codes, actions, noise, inv_m_global

    def d_grads(self, first):
        if first:
            self.sm.g_forward(self.codes, self.actions, self.noise)
        self._dg = self.sm.d_grads(self.inv_m)
        return _flat(self._dg)

    def apply_d(self, grad):
        self.sm.apply_d(_unflat(grad, self._dg))

    def g_grads(self):
        self._gg = self.sm.g_grads(self.inv_m)
        return _flat(self._gg)

    def apply_g(self, grad):
        self.sm.apply_g(_unflat(grad, self._gg))


def _worker(rank, world, port, dsteps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from ndivplanning_amd import dp
    from oracle import gan_oracle as O
    r, w = dp.init_process_group(None)
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    batch, k, steps = 4, 3, 2
    codes, actions, noise = O.synthetic_batch(5, batch, k, steps=steps)
    flat = codes.shape[0]
    lo, hi = dp.shard_bounds(flat, rank, world)
    g, d = O.init_params(0, 2)
    be = OracleBackend(O, g, d, codes[lo:hi], actions[lo:hi], None, 1.0 / (flat * k))
    shares = []
    for s in range(steps):
        be.noise = noise[s][lo:hi]
        dp.run_step(be, dp.sum_all_reduce(), discrim_steps=dsteps)
        o = be.sm.out
        shares.append(dp.reduce_loss_shares([o["d_loss"].item(), o["g_loss"].item(), o["pair_div"].item()]))
    torch.save({"g": _flat(be.sm.g), "d": _flat(be.sm.d), "losses": shares}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dsteps", [1, 2])
def test_two_ranks_equal_global_batch(tmp_path, dsteps):
    sys.path.insert(0, ROOT)
    from oracle import gan_oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), dsteps, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    # replicas stay identical
    assert torch.equal(res[0]["g"], res[1]["g"]) and torch.equal(res[0]["d"], res[1]["d"])
    # single process, whole batch
    torch.set_num_threads(1)
    batch, k, steps = 4, 3, 2
    codes, actions, noise = O.synthetic_batch(5, batch, k, steps=steps)
    g, d = O.init_params(0, 2)
    sm = O.StepMath(g, d)
    for s in range(steps):
        out = sm.step(codes, actions, noise[s], discrim_steps=dsteps)
        want = [out["d_loss"].item(), out["g_loss"].item(), out["pair_div"].item()]
        got = res[0]["losses"][s]
        assert abs(got[0] - want[0]) < 1e-5 and abs(got[1] - want[1]) < 1e-5
        assert abs(got[2] - want[2]) <= (1e-4 if s == 0 else 2e-2) * max(1.0, abs(want[2]))
    # parameters: same update rule on summed gradients; differences are Adam's +-lr moves on
    # gradient elements at the fp32 noise floor (summation order differs between 1 and 2 shards)
    for name, a, b in (("G", res[0]["g"], _flat(sm.g)), ("D", res[0]["d"], _flat(sm.d))):
        err = (a - b).abs()
        assert err.max() <= 2.5 * 2e-4 * steps * dsteps, name
        assert (err > 1e-4).float().mean() <= 0.06, name


def _mean_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from ndivplanning_amd import dp
    dp.init_process_group(None)
    lo, hi = dp.shard_bounds(8, rank, world)
    grad = torch.arange(6, dtype=torch.float32) * (rank + 1) + lo          # what a rank's backward left in its flat vector
    dp.mean_all_reduce(world)(grad)
    torch.save({"grad": grad, "bounds": (lo, hi)}, os.path.join(out_dir, "mean%d.pt" % rank))
    dist.destroy_process_group()


def test_mean_all_reduce_of_the_forward_model_recipe(tmp_path):
    """The forward model's data-parallel exchange (train_forward_model.py mirror): every rank ends up with the MEAN of the
    ranks' flat gradients, in place."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_mean_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), "mean%d.pt" % r)) for r in range(2))
    want = (torch.arange(6, dtype=torch.float32) * 1 + 0 + torch.arange(6, dtype=torch.float32) * 2 + 4) / 2
    assert torch.equal(a["grad"], want) and torch.equal(b["grad"], want)
    assert a["bounds"] == (0, 4) and b["bounds"] == (4, 8)


def _lockstep_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from ndivplanning_amd import dp
    dp.init_process_group(None)
    g, d = torch.linspace(-1, 1, 83780), torch.linspace(0, 3, 58305)
    same = dp.replicas_bit_identical([g, d])
    # one rank's replica differs in the LAST bit of one parameter: a sum that arrived wrong once
    if rank == 1:
        d.view(torch.int32)[12345] ^= 1
    differ = dp.replicas_bit_identical([g, d])
    raised = None
    try:
        dp.assert_replicas_identical([g, d], "p2p", "end of epoch 0")
    except dp.ReplicaDivergence as exc:
        raised = str(exc)
    # two values swapped on one rank: the plain sum of the bits cannot see it, the position-weighted one must
    if rank == 1:
        d.view(torch.int32)[12345] ^= 1
        g[[5, 9]] = g[[9, 5]]
    swapped = dp.replicas_bit_identical([g, d])
    torch.save({"same": same, "differ": differ, "raised": raised, "swapped": swapped}, os.path.join(out_dir, "ls%d.pt" % rank))
    dist.destroy_process_group()


def test_replica_checksums_catch_a_one_bit_divergence_on_every_rank(tmp_path):
    """What train_gan runs after the first launch and once per epoch of a data-parallel run (and bench.py after timing):
    identical replicas pass; one flipped bit or two swapped values on one rank fail on BOTH ranks, and the error names
    the exchange."""
    mp.spawn(_lockstep_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), "ls%d.pt" % r))
        assert res["same"] is True and res["differ"] is False and res["swapped"] is False
        assert res["raised"] is not None and "'p2p'" in res["raised"] and "end of epoch 0" in res["raised"]
