"""Diagnostic (GPU box): 2 ranks x 2 images with cross-rank BatchNorm statistics vs 1 process x 4 images: per-tensor gradient error."""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.multiprocessing as mp


def data(n):
    gen = torch.Generator().manual_seed(7)
    cur = torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1
    fut = torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1
    act = torch.rand(n, 4, generator=gen) * 2 - 1
    return cur, fut, act


def make(n, world=1):
    from ndivplanning_amd.forward_trainer import ForwardModelTrainer
    from ndivplanning_amd.models import forward_encoder as FE
    from oracle import forward_model_oracle as FO
    model = FE.ForwardAutoencoder()
    model.load_state_dict(FO.init_forward_model_state(3))
    return ForwardModelTrainer(model.to("cuda:0").train(), batch=n, sync_batchnorm_world=world)


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), NDP_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from ndivplanning_amd import dp
    dp.init_process_group("cuda:0")
    tr = make(2, world)
    cur, fut, act = data(4)
    sl = slice(2 * rank, 2 * rank + 2)
    tr.grads(cur[sl].to("cuda:0"), fut[sl].to("cuda:0"), act[sl].to("cuda:0"))
    g = tr.grad.clone()
    dp.mean_all_reduce(world)(g)
    torch.save({"grad": g.cpu(), "loss": tr.loss.item(), "stats": tr.stats.cpu(), "calls": tr.stat_sync.calls}, os.path.join(out, "r%d.pt" % rank))
    tr.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    out = sys.argv[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    res = [torch.load(os.path.join(out, "r%d.pt" % r)) for r in range(2)]
    tr = make(4)
    cur, fut, act = data(4)
    tr.grads(cur.to("cuda:0"), fut.to("cuda:0"), act.to("cuda:0"))
    from ndivplanning_amd.models import forward_encoder as FE
    print("callbacks per rank", res[0]["calls"], "loss single", tr.loss.item(), "ranks", res[0]["loss"], res[1]["loss"], "mean", (res[0]["loss"] + res[1]["loss"]) / 2)
    print("stats max abs diff", float((tr.stats.cpu() - res[0]["stats"]).abs().max()), "ranks equal", torch.equal(res[0]["stats"], res[1]["stats"]))
    want = FE.unpack_vector(tr.grad, tr.model)
    got = FE.unpack_vector(res[0]["grad"].to("cuda:0"), tr.model)
    for k in want:
        a, b = got[k].double().cpu(), want[k].double().cpu()
        print("%-34s rel %.2e   |want| %.2e" % (k, float((a - b).norm() / b.norm().clamp_min(1e-30)), float(b.norm())))
