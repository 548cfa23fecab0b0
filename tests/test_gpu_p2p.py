"""GPU tests of the in-kernel peer-to-peer gradient exchange (include/ndp.h, ndp_p2p_*; dp.P2PExchange).

The GPU box has ONE card, so the ranks are separate PROCESSES sharing cuda:0: the regions are still exchanged with
hipIpc and the kernels of the ranks still hand-shake through flags while running concurrently -- the protocol
(push, release, bounded wait, rank-ordered sum, parity double-buffering) is what is tested; the xGMI transport
between different cards is not, which is why dp.make_exchange self-checks on the node it runs on."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import gan_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH_GLOBAL, K, NZ, STEPS = 12, 6, 2, 5         # FLAT_global = 84: splits over 2 and 3 ranks


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_trainer(flat, flat_global, p2p, nslots=1, use_graph=True, dsteps=1):
    from ndivplanning_amd.models.gan import Decoder, Discriminator
    from ndivplanning_amd.trainer import GanTrainer
    g0, d0 = O.init_params(0, NZ)
    dec, dis = Decoder(noise_dim=NZ).cuda(), Discriminator().cuda()
    dec.load_state_dict(g0)
    dis.load_state_dict(d0)
    return GanTrainer(dec, dis, flat=flat, num_sample=K, flat_global=flat_global, p2p=p2p, use_graph=use_graph,
                      steps_per_launch=nslots, discrim_steps=dsteps)


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", NDP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from ndivplanning_amd import dp
    torch.cuda.set_device(0)
    dp.init_process_group("cuda:0")
    res = {}
    p2p = dp.P2PExchange("cuda:0", timeout_ms=20000)
    res["self_check"] = p2p.self_check(trials=4, check_timeout_ms=20000) and p2p.shared_device
    # the exchange alone, odd sizes, both nets, many consecutive exchanges (parity double-buffering)
    word = torch.zeros(4, dtype=torch.int32, device="cuda")
    exact = True
    for i in range(1, 41):
        n = (58305, 83780, 1, 257)[i % 4]
        net = 1 if n > 58305 else i % 2
        x = torch.full((n,), float(rank + 1), device="cuda") * i + torch.arange(n, device="cuda") % 7
        word.fill_(i)
        got = p2p.all_reduce(x, word, net=net)
        want = sum(float(r + 1) for r in range(world)) * i + world * (torch.arange(n, device="cuda") % 7)
        exact = exact and bool(torch.equal(got, want.float()))
    res["exact"] = exact and p2p.status() == 0
    p2p.reset()

    codes, actions, noise = O.synthetic_batch(11, BATCH_GLOBAL, K, NZ, steps=STEPS)
    flat_g = codes.shape[0]
    lo, hi = dp.shard_bounds(flat_g, rank, world)
    t = _make_trainer(hi - lo, flat_g, p2p, use_graph=False)
    c, a = codes[lo:hi].cuda(), actions[lo:hi].cuda()
    t.step(c, a, noise[0, lo:hi].cuda())
    torch.cuda.synchronize()
    res["d_grad1"], res["g_grad1"] = t.d_grad.cpu(), t.g_grad.cpu()
    res["losses1"] = dp.reduce_loss_shares(t.losses())
    for s in range(1, STEPS):
        t.step(c, a, noise[s, lo:hi].cuda())
    torch.cuda.synchronize()
    res["g_eager"], res["d_eager"] = t.g_flat.detach().cpu().clone(), t.d_flat.detach().cpu().clone()
    p2p.check()

    # the same 5 steps as ONE graph replay (5 slots) on a fresh exchange state: bit-identical to eager
    p2p.reset()
    t2 = _make_trainer(hi - lo, flat_g, p2p, nslots=STEPS)
    t2.step_many(c.expand(STEPS, -1, -1).contiguous(), a.expand(STEPS, -1, -1).contiguous(), noise[:, lo:hi].cuda())
    torch.cuda.synchronize()
    res["g_graph"], res["d_graph"] = t2.g_flat.detach().cpu().clone(), t2.d_flat.detach().cpu().clone()
    p2p.check()
    res["status"] = p2p.status()
    torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    del t, t2
    p2p.close()
    dist.destroy_process_group()


# (2 and 3 ranks.  The hand-shake needs every rank's reduce kernel resident at the same time; ranks that are
# processes on ONE GPU depend on how the hardware scheduler interleaves their queues for that -- round 1 saw a
# bounded-wait time-out in 1 run of 3 with FOUR ranks and kept no record of it.  That precondition is now explicit:
# dp.make_exchange refuses the exchange for ranks that share a GPU unless it is forced, as it is here, and a
# time-out leaves a diagnostic record (test_p2p_wait_is_bounded_when_a_peer_never_pushes).)
@pytest.mark.parametrize("world", [2, 3])
def test_p2p_exchange_ranks_as_processes_on_one_gpu(tmp_path, world):
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for r in res:
        assert r["self_check"] and r["exact"] and r["status"] == 0
    # replicas stay bit-identical (every rank sums in rank order), eager and graph give the same bits
    for r in res[1:]:
        for key in ("d_grad1", "g_grad1", "g_eager", "d_eager", "g_graph", "d_graph"):
            assert torch.equal(r[key], res[0][key]), key
    assert torch.equal(res[0]["g_eager"], res[0]["g_graph"]) and torch.equal(res[0]["d_eager"], res[0]["d_graph"])

    # against one process training the whole global batch
    codes, actions, noise = O.synthetic_batch(11, BATCH_GLOBAL, K, NZ, steps=STEPS)
    t = _make_trainer(codes.shape[0], codes.shape[0], None, use_graph=False)
    t.step(codes.cuda(), actions.cuda(), noise[0].cuda())
    torch.cuda.synchronize()
    d_ref, g_ref = t.d_grad.cpu(), t.g_grad.cpu()
    # D gradient: same mathematics, different summation split -> fp32 noise only
    assert (res[0]["d_grad1"] - d_ref).abs().max() <= 1e-5 * d_ref.abs().max() + 1e-7
    # G gradient runs through the updated D, whose Adam step turns noise-level gradient differences into
    # +-lr moves on a few parameters (tests/test_gpu_parity.py discusses it): looser bound
    assert (res[0]["g_grad1"] - g_ref).abs().max() <= 1e-2 * g_ref.abs().max()
    want = t.losses()
    got = res[0]["losses1"]
    assert abs(got[0] - want[0]) <= 1e-5 and abs(got[1] - want[1]) <= 1e-4
    assert abs(got[2] - want[2]) <= 1e-4 * max(1.0, abs(want[2]))


def test_p2p_wait_is_bounded_when_a_peer_never_pushes(tmp_path):
    """A rank whose peer does not take part times out (status word), it does not hang."""
    mp.spawn(_timeout_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "t%d.pt" % r)) for r in range(2)]
    assert res[0]["status"] == 2          # rank 0 waited for rank 1 (1 + 1)
    assert res[0]["seconds"] < 5.0
    assert res[1]["status"] == 0 and res[1]["diag"] is None
    # the record the first waiter left: which peer, which net, which exchange number it expected and what it saw
    d = res[0]["diag"]
    assert d["peer"] == 1 and d["net"] == "D" and d["expected_step"] == 1 and d["flag_seen"] == 0
    assert 250 <= d["waited_ms"] <= 2000
    assert "timed out waiting for rank 1" in res[0]["poll"] and "share a GPU" in res[0]["poll"]


def _guard_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", NDP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("NDP_DP_EXCHANGE", None)
    import torch.distributed as dist
    from ndivplanning_amd import dp
    torch.cuda.set_device(0)
    dp.init_process_group("cuda:0")
    logs = []
    p2p, reduce_fn, name = dp.make_exchange("cuda:0", world, log=logs.append)
    torch.save({"p2p": p2p is not None, "reduce": reduce_fn is not None, "name": name, "logs": logs},
               os.path.join(out_dir, "g%d.pt" % rank))
    dist.destroy_process_group()


def test_make_exchange_refuses_the_in_kernel_exchange_for_ranks_that_share_a_gpu(tmp_path):
    """The hand-shake's precondition -- every rank's reduce kernel resident at once -- is only guaranteed with a GPU
    per rank: `make_exchange` compares the ranks' PCI bus ids and, unless NDP_DP_EXCHANGE=p2p forces it, selects the
    collective on every rank alike."""
    mp.spawn(_guard_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "g%d.pt" % r)) for r in range(2)]
    for r in res:
        assert not r["p2p"] and r["reduce"] and r["name"] == "rccl"
        assert any("share a GPU" in m for m in r["logs"])


def _timeout_main(rank, world, port, out_dir):
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", NDP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from ndivplanning_amd import dp
    torch.cuda.set_device(0)
    dp.init_process_group("cuda:0")
    p2p = dp.P2PExchange("cuda:0", timeout_ms=300)
    word = torch.ones(4, dtype=torch.int32, device="cuda")
    t0 = time.time()
    if rank == 0:
        p2p.all_reduce(torch.ones(1000, device="cuda"), word, net=0)     # rank 1 never answers
        p2p.all_reduce(torch.ones(1000, device="cuda"), word + 1, net=0)  # sticky status: no second wait
    torch.cuda.synchronize()
    res = {"status": p2p.status(), "seconds": time.time() - t0, "diag": p2p.diagnostics()}
    if rank == 0:
        try:
            p2p.poll()                       # first call only enqueues the copy of the status word ...
            torch.cuda.synchronize()
            p2p.poll()                       # ... the next one sees it and raises, with the record
            res["poll"] = "no error"
        except RuntimeError as exc:
            res["poll"] = str(exc)
    torch.save(res, os.path.join(out_dir, "t%d.pt" % rank))
    p2p.close()
    dist.destroy_process_group()


def _dsteps_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", NDP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from ndivplanning_amd import dp
    torch.cuda.set_device(0)
    dp.init_process_group("cuda:0")
    p2p = dp.P2PExchange("cuda:0", timeout_ms=20000)
    codes, actions, noise = O.synthetic_batch(12, BATCH_GLOBAL, K, NZ, steps=3)
    lo, hi = dp.shard_bounds(codes.shape[0], rank, world)
    t = _make_trainer(hi - lo, codes.shape[0], p2p, dsteps=2)          # graph: 2 D exchanges + 1 G exchange per step
    hist = []
    for s_ in range(3):
        t.step(codes[lo:hi].cuda(), actions[lo:hi].cuda(), noise[s_, lo:hi].cuda())
        hist.append(dp.reduce_loss_shares(t.losses()))
    torch.cuda.synchronize()
    torch.save({"g": t.g_flat.detach().cpu(), "d": t.d_flat.detach().cpu(), "hist": hist, "status": p2p.status(),
                "d_steps": int(t.d_step[0]), "g_steps": int(t.g_step[0])}, os.path.join(out_dir, "d%d.pt" % rank))
    del t
    p2p.close()
    dist.destroy_process_group()


def test_p2p_two_discriminator_steps_per_iteration(tmp_path):
    """discrim_steps_per_gen = 2 (train_gan.py:172): the repeat D step runs through k_d + k_wgrad + the exchanging
    reduce kernel as well; D's exchange number advances twice per iteration, G's once."""
    mp.spawn(_dsteps_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "d%d.pt" % r)) for r in range(2)]
    assert res[0]["status"] == 0 and res[1]["status"] == 0
    assert res[0]["d_steps"] == 6 and res[0]["g_steps"] == 3
    assert torch.equal(res[0]["g"], res[1]["g"]) and torch.equal(res[0]["d"], res[1]["d"])
    assert res[0]["hist"] == res[1]["hist"]
    codes, actions, noise = O.synthetic_batch(12, BATCH_GLOBAL, K, NZ, steps=3)
    t = _make_trainer(codes.shape[0], codes.shape[0], None, dsteps=2)
    t.step(codes.cuda(), actions.cuda(), noise[0].cuda())
    want = t.losses()
    got = res[0]["hist"][0]
    assert abs(got[0] - want[0]) <= 1e-5 and abs(got[1] - want[1]) <= 1e-4
