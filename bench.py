#!/usr/bin/env python3
"""bench.py -- GAN train steps/sec of the HIP path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one iteration of the reference's train_gan.py loop body (train_gan.py:159-203:
one D update + one G update) on one synthetic codes-mode batch, BASELINE config 2:
B = 64 trajectories, traj_len = 8 (FLAT = 448 rows), K = 6 samples (M = 2,688 rows), nz = 2.
Inputs (codes, actions) are resident in HBM; the K noise samples are drawn per step on the
device inside the timed region; nothing is skipped (both Adam updates, all three losses).

N > 1 is weak scaling: every rank trains its own 64-trajectory shard of a global batch of
64*N (gradients summed by one RCCL all-reduce per network per step), so `value` counts
batch-64 steps: N * iterations/s.

Rank 0 prints one JSON line (contract in the task statement) with two extra objects:
  roofline     the dominant kernel of the step vs the fp32-MFMA peak (157.3 TFLOP/s):
               algorithmic FLOPs of that kernel per launch / its average duration, measured
               with HIP events on the launch stream (ndp_timing_*), plus every kernel's share;
  cpu_baseline the oracle's restated reference loop (torch CPU fp32) timed on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0
# HBM bytes per launch of the step's kernels at the DEFAULT workload (B=64, K=6), from separate
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (profiles/r01_v7_pmc_hbm.csv), with the
# gfx950 correction of MI355X_MICROARCH.md section HBM: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
# bench.py cannot read PMC counters itself; the figure is reported only for the workload it was
# measured on.
PMC_HBM_BYTES_DEFAULT = {"k_phase_a": 33656488, "k_phase_b": 18603529, "k_wgrad[D]": 34154522,
                         "k_wgrad[G]": 21469747}

# algorithmic MACs per M-row of each kernel (629,760 per row per step)
G_FWD = 128 * 258 + 64 * 128 + 128 * 64 + 256 * 128 + 4 * 256           # 83,200
D_FWD = 64 * 260 + 128 * 64 + 256 * 128 + 256                            # 57,856
D_DGRAD = 256 + 256 * 128 + 128 * 64                                     # 41,216
G_DGRAD = 4 * 256 + 256 * 128 + 128 * 64 + 64 * 128                      # 50,176
KERNEL_MACS_PER_ROW = {
    "k_phase_a": G_FWD + 2 * (D_FWD + D_DGRAD),          # G forward + D(real), D(fake) forward/backward data path
    "k_wgrad[D]": 2 * D_FWD,
    "k_phase_b": D_FWD + D_DGRAD + 4 * 64 + G_DGRAD,     # D' forward/backward to action_hat + G backward data path
    "k_wgrad[G]": G_FWD,
}
assert sum(KERNEL_MACS_PER_ROW.values()) == 629760
# kernels of the non-fused entry points (repeat D steps, module API); not part of the default step
KERNEL_MACS_PER_ROW.update({"k_g_fwd": G_FWD, "k_d[2 pass fwd+bwd]": 2 * (D_FWD + D_DGRAD),
                            "k_d[fwd+bwd]": D_FWD + D_DGRAD + 4 * 64, "k_g_bwd": G_DGRAD})


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=64, help="trajectories per GPU (BASELINE config 2: 64)")
    ap.add_argument("--num-sample", type=int, default=6)
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--steps-per-launch", type=int, default=16,
                    help="iterations captured per HIP graph (single GPU; each iteration has its own input slot)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel code path (non-fused Adam + RCCL all-reduce) even with one rank")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def cpu_baseline(batch, k, nz, seconds):
    """The reference arithmetic (oracle.AutogradTrainer: the restated train_gan.py loop on
    torch CPU fp32) timed on this host's cores: a bounded sample of the same workload."""
    from oracle import gan_oracle as O
    g, d = O.init_params(0, nz)
    codes, actions, noise = O.synthetic_batch(0, batch, k, nz, steps=8)
    tr = O.AutogradTrainer(g, d)
    # torch's default (= all hardware threads) oversubscribes these small GEMMs badly; probe a
    # few thread counts for ~1 s each and report the best one -- the fair baseline
    default_threads = torch.get_num_threads()
    best = (0.0, default_threads)
    for cand in sorted({1, 4, 8, 16, 32, default_threads}):
        if cand > default_threads:
            continue
        torch.set_num_threads(cand)
        tr.step(codes, actions, noise[0])
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            tr.step(codes, actions, noise[n % 8])
            n += 1
        rate = n / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, cand)
    threads = best[1]
    torch.set_num_threads(threads)
    for s in range(3):
        tr.step(codes, actions, noise[s % 8])
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(codes, actions, noise[n % 8])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 5000:
            break
    return {"value": round(n / dt, 3), "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "%d steps of the same B=%d,K=%d codes-mode workload in %.1f s (torch %s CPU fp32; best of "
                      "{1,4,8,16,32,%d} threads = %d)" % (n, batch, k, dt, torch.__version__, default_threads, threads)}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NDP_BENCH_ONE_GPU") == "1":
        # rehearsal of the N > 1 path on a one-GPU box: all ranks share cuda:0 (needs NDP_DIST_BACKEND=gloo,
        # RCCL refuses two ranks on one device); the number it prints is not a measurement
        local_rank = 0
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from ndivplanning_amd import dp
    reduce_fn, p2p, exchange = None, None, "none"
    if world > 1 or args.force_dp:
        dp.init_process_group(dev, force=args.force_dp)     # RCCL; keeps its banner off stdout
        # gradient exchange: in-kernel peer-to-peer (hipIpc over xGMI) if it passes its self-check on
        # this node, RCCL all-reduce between the phases otherwise (NDP_DP_EXCHANGE=rccl|p2p forces one)
        if world > 1:
            p2p, reduce_fn, exchange = dp.make_exchange(
                dev, world, log=(lambda m_: print("[bench] " + m_, file=sys.stderr)) if rank == 0 else None)
        else:
            reduce_fn, exchange = dp.sum_all_reduce(), "rccl"

    from ndivplanning_amd import _capi
    from ndivplanning_amd.models.gan import Decoder, Discriminator
    from ndivplanning_amd.trainer import GanTrainer
    from oracle import gan_oracle as O   # inputs + the step-0 parity figure + cpu_baseline only

    batch, k, nz, traj = args.batch, args.num_sample, 2, 8
    flat = batch * (traj - 1)
    m = flat * k
    g, d = O.init_params(0, nz)
    dec, dis = Decoder(nz), Discriminator()
    dec.load_state_dict(g)
    dis.load_state_dict(d)
    dec, dis = dec.to(dev), dis.to(dev)
    codes, actions, noise = O.synthetic_batch(1000 + rank, batch, k, nz, steps=1)
    spl = args.steps_per_launch if (reduce_fn is None and not args.no_graph) else 1

    def make_trainer():
        dec.load_state_dict(g)
        dis.load_state_dict(d)
        return GanTrainer(dec, dis, flat=flat, num_sample=k, flat_global=flat * world, reduce_fn=reduce_fn, p2p=p2p,
                          use_graph=not args.no_graph, noise_seed=rank, steps_per_launch=spl)
    tr = make_trainer()

    # step-0 parity figure (outside the timed region): NDiv / losses vs the oracle on rank 0's shard
    parity = None
    if world == 1:
        ref = O.StepMath({n_: v.clone() for n_, v in g.items()}, {n_: v.clone() for n_, v in d.items()})
        out = ref.step(codes, actions, noise[0])
        tr.step(codes.to(dev), actions.to(dev), noise[0].to(dev))
        dl, gl, pd = tr.losses()
        parity = {"ndiv_rel_err": abs(pd - out["pair_div"].item()) / max(1.0, abs(out["pair_div"].item())),
                  "d_loss_abs_err": abs(dl - out["d_loss"].item()), "g_loss_abs_err": abs(gl - out["g_loss"].item()),
                  "action_hat_max_abs_err": (tr.action_hat[:m].cpu() - out["action_hat"]).abs().max().item()}

    def fill_slots():
        # every input slot holds its own resident synthetic batch
        tr.codes.copy_(codes)
        tr.actions.copy_(actions)
        for slot in range(1, spl):
            c_, a_, _ = O.synthetic_batch(2000 + 17 * slot + rank, batch, k, nz, steps=1)
            tr.codes_slots[slot].copy_(c_)
            tr.actions_slots[slot].copy_(a_)
    fill_slots()

    def run_steps(n):
        done = 0
        while spl > 1 and n - done >= spl:
            tr.step_many()             # spl iterations, one graph replay
            done += spl
        for _ in range(n - done):
            tr.step()                  # device noise, resident inputs

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    run_steps(args.warmup)
    if p2p is not None:
        # a wait that timed out during warm-up (status word) means the exchange does not work on this
        # node although its self-check passed: fall back to the RCCL path rather than time garbage
        bad = torch.tensor([p2p.status()], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) != 0:
            print("[bench] peer-to-peer exchange timed out in warm-up; falling back to RCCL", file=sys.stderr)
            del tr
            p2p.close()
            p2p, reduce_fn, exchange, spl = None, dp.sum_all_reduce(), "rccl (p2p timed out in warm-up)", 1
            tr = make_trainer()
            fill_slots()
            barrier()
            run_steps(args.warmup)
    run_steps(spl + 1)                 # make sure both graphs exist before timing
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = tr.losses()
    hip_graph = bool(tr.use_graph)

    # per-kernel durations: HIP events around every launch, eager launches of the same step
    # (every rank steps -- the exchange needs all of them -- rank 0 records)
    kernels = {}
    saved = tr.use_graph
    tr.use_graph = False
    if rank == 0:
        _capi.timing_enable(True)
    reps = 50
    for _ in range(reps):
        tr.step()
    torch.cuda.synchronize(dev)
    if rank == 0:
        timed = _capi.timing_collect()
        _capi.timing_enable(False)
        for name, (ms, cnt) in timed.items():
            us = 1e3 * ms / max(cnt, 1)
            macs = KERNEL_MACS_PER_ROW.get(name)
            kernels[name] = {"avg_us": round(us, 3), "launches_per_step": cnt / reps,
                             "tflops": round(2.0 * macs * m / (us * 1e-6) / 1e12, 3) if macs else None}
    tr.use_graph = saved

    replicas_identical = None
    if world > 1:
        # the replicas must have stayed bit-identical: compare a checksum of the parameter bits
        bits = torch.cat([tr.g_flat.detach(), tr.d_flat.detach()]).view(torch.int32).to(torch.int64)
        mine = torch.stack([bits.sum(), (bits * torch.arange(1, bits.numel() + 1, device=dev)).sum()])
        lo_, hi_ = mine.clone(), mine.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo_, hi_))
        if p2p is not None:
            p2p.check()                # raises if any wait timed out
            del tr
            p2p.close()
    if rank != 0:
        dist.destroy_process_group()
        return

    iters_per_s = args.steps / elapsed
    value = iters_per_s * world
    dom = max((n_ for n_ in kernels if KERNEL_MACS_PER_ROW.get(n_)), key=lambda n_: kernels[n_]["avg_us"] * kernels[n_]["launches_per_step"])
    dom_flops = 2.0 * KERNEL_MACS_PER_ROW[dom] * m
    achieved = dom_flops / (kernels[dom]["avg_us"] * 1e-6) / 1e12
    step_flops = 2.0 * 629760 * m
    result = {
        "metric": "gan_train_steps_per_sec_traj8_batch64", "value": round(value, 2), "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: train_gan.py step (D update + G update), codes mode, "
                               "batch=%d trajectories per GPU, traj_len=8, num_sample=%d, noise_dim=2" % (batch, k),
                   "rows_per_gpu": m, "global_batch": batch * world, "parallelism": "dp%d" % world,
                   "trajectories_per_sec": round(iters_per_s * batch * world, 1),
                   "hip_graph": hip_graph, "steps_per_graph_launch": spl if hip_graph else 0,
                   "gradient_exchange": exchange, "all_reduce_us": dict(dp.last_exchange_report) or None,
                   "replicas_bit_identical": replicas_identical, "last_losses": {"D": losses[0], "G": losses[1], "ndiv": losses[2]}},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": MFMA_F32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": PMC_HBM_BYTES_DEFAULT.get(dom) if (batch, k) == (64, 6) else None,
                     "traffic_source": "profiles/r01_v7_pmc_hbm.csv (rocprofv3 --pmc, bytes per launch)",
                     "algorithmic_flops_per_launch": dom_flops,
                     "whole_step": {"flops": step_flops,
                                    "tflops": round(step_flops * iters_per_s / 1e12, 3),
                                    "frac": round(step_flops * iters_per_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                     "kernels": kernels},
    }
    if parity is not None:
        result["parity_step0"] = {k_: float("%.3e" % v) for k_, v in parity.items()}
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(batch, k, nz, args.cpu_seconds)
        result["cpu_baseline"]["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
