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
64*N (gradients summed over ranks once per network per step), so `value` counts batch-64
steps: N * iterations/s.

Timing: W warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides,
maximum over ranks.  When K steps take less than ~50 ms the bracket is repeated R times back to
back and the MEDIAN repetition is reported (`config.repeats`, all repetitions in
`config.repeat_ms_per_step`): a 20-step run is 1.9 ms of GPU time, one sample of which says little.

Rank 0 prints one JSON line (contract in the task statement) with extra objects:
  roofline     the dominant kernel of the step vs the fp32-MFMA peak (157.3 TFLOP/s):
               algorithmic FLOPs of that kernel per launch / its average duration, measured
               with HIP events on the launch stream (ndp_timing_*), plus every kernel's share;
  cpu_baseline the oracle's restated reference loop (torch CPU fp32) timed on this host (N = 1);
  h2d_per_launch   the same workload with a FRESH batch uploaded from pinned host memory for every
               step (the reference uploads per step, train_gan.py:119-124), overlapped with the
               previous graph launch (N = 1);
  config4      BASELINE configs[3]: image-conditioned step at B = 128 -- frozen encoder over the
               1,024 unique frames + the fused step at M = 5,376 (N = 1);
  large_m      the config-5 per-GPU shard (B = 128, K = 32) and B = 1,024 / K = 6: whole-step
               fraction of the fp32-MFMA peak where the matrix pipe, not launch latency, is the story;
  forward_model    SURVEY.md section 8 row f4: one iteration of train_forward_model.py (U-Net forward, MSE,
               backward, Adam) at batch 8 and 32, with its kernel-time shares and the oracle timed on the CPU (N = 1);
  strong_config3, config5_shard   (N > 1) global batch 256 split over the ranks, and the config-5 shard
               per rank, with the gradient exchange that ran and both exchanges timed;
  forward_model_dp   (N > 1) the forward-model iteration at 8 images per rank with its gradient all-reduce: per bucket,
               overlapped with the backward pass, and as one collective; exposed_all_reduce_ms for both.
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# read by HSA when the runtime initialises (first torch.cuda call): dmabuf IPC for hipIpc / RCCL
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0
# HBM bytes per launch of the step's kernels at the DEFAULT workload (B=64, K=6), from separate
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (profiles/*_pmc_hbm.csv), with the
# gfx950 correction of MI355X_MICROARCH.md section HBM: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
# bench.py cannot read PMC counters itself; the figure is reported only for the workload it was
# measured on.
PMC_HBM_BYTES_DEFAULT = {"k_phase_a": 25417455, "k_phase_b": 18645523, "k_wgrad[D]": 18803484,
                         "k_wgrad[G]": 22566634}
PMC_SOURCE = "profiles/r03_final_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)"
# matrix-pipe utilisation of the same kernels from SQ counters (SQ_VALU_MFMA_BUSY_CYCLES over the launch's SIMD-cycles at
# 2.4 GHz; profiles/README.md, round 3): reported beside the event-timed fraction, for the default workload only
PMC_MFMA_BUSY_DEFAULT = {"k_phase_a": 0.2547, "k_phase_b": 0.2280, "k_wgrad[D]": 0.1402, "k_wgrad[G]": 0.1876}
PMC_MFMA_SOURCE = "profiles/r03_cfg2_pmc_sq.csv (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ...; also wait / LDS / L2 shares)"

# algorithmic MACs per M-row of each kernel (629,760 per row per step)
G_FWD = 128 * 258 + 64 * 128 + 128 * 64 + 256 * 128 + 4 * 256           # 83,200
D_FWD = 64 * 260 + 128 * 64 + 256 * 128 + 256                            # 57,856
D_DGRAD = 256 + 256 * 128 + 128 * 64                                     # 41,216
G_DGRAD = 4 * 256 + 256 * 128 + 128 * 64 + 64 * 128                      # 50,176
MACS_PER_ROW_STEP = 629760       # the reference's step, autograd-minimal, per M-row


def kernel_macs(k):
    """MACs per M-row that each kernel EXECUTES.  `real` = share of the D step's real pass that runs: the reference
    feeds D K identical copies of every real (action, code) pair (train_gan.py:140-156); k_phase_a / k_wgrad[D] run it
    on the FLAT distinct rows, weighted K -- 1/K of the rows.  With real = 1 the table is SURVEY.md's 629,760."""
    real = 1.0 / k
    macs = {
        "k_phase_a": G_FWD + (1 + real) * (D_FWD + D_DGRAD),  # G forward + D(fake), D(real) forward/backward data path
        "k_wgrad[D]": (1 + real) * D_FWD,
        "k_phase_b": D_FWD + D_DGRAD + 4 * 64 + G_DGRAD,      # D' forward/backward to action_hat + G backward data path
        "k_wgrad[G]": G_FWD,
    }
    executed = sum(macs.values())
    # kernels of the non-fused entry points (repeat D steps, module API); not part of the default step
    macs.update({"k_g_fwd": G_FWD, "k_d[2 pass fwd+bwd]": 2 * (D_FWD + D_DGRAD),
                 "k_d[fwd+bwd]": D_FWD + D_DGRAD + 4 * 64, "k_g_bwd": G_DGRAD})
    return macs, executed


assert abs(kernel_macs(1)[1] - MACS_PER_ROW_STEP) < 1e-6
ENCODER_FLOP_PER_IMAGE = 622.3e6                          # SURVEY.md section 8d (311.2 M MAC)
TRAJ, NZ = 8, 2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=64, help="trajectories per GPU (BASELINE config 2: 64)")
    ap.add_argument("--num-sample", type=int, default=6)
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--steps-per-launch", type=int, default=16,
                    help="iterations captured per HIP graph (each iteration has its own input slot)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (profiling runs)")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel code path (non-fused Adam + RCCL all-reduce) even with one rank")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="repeat the K-step bracket until about this much GPU time has been sampled; report the median")
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
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or default_threads
        logical = psutil.cpu_count(logical=True) or default_threads
    except Exception:                                     # noqa: BLE001
        physical = logical = os.cpu_count() or default_threads
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
    torch.set_num_threads(default_threads)
    return {"value": round(n / dt, 3), "unit": "steps/s", "cores": physical, "threads": threads,
            "logical_cpus": logical, "kind": "port",
            "sample": "%d steps of the same B=%d,K=%d codes-mode workload in %.1f s (torch %s CPU fp32; host has %d "
                      "physical cores / %d hardware threads; best of {1,4,8,16,32,%d} torch threads = %d -- more "
                      "threads are slower on these small GEMMs)"
                      % (n, batch, k, dt, torch.__version__, physical, logical, default_threads, threads)}


class Bench:
    """One process' view of the run: rank, device, the gradient exchange, the seeded networks."""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("NDP_BENCH_ONE_GPU") == "1":
            # rehearsal of the N > 1 path on a one-GPU box: all ranks share cuda:0 (needs NDP_DIST_BACKEND=gloo,
            # RCCL refuses two ranks on one device); the number it prints is not a measurement
            local_rank = 0
        if args.gpus != self.world and self.world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus)
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path exists)"
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        import torch.distributed as dist
        from ndivplanning_amd import dp
        self.dist, self.dp = dist, dp
        self.p2p, self.reduce_fn, self.exchange = None, None, "none"
        if self.world > 1 or args.force_dp:
            dp.init_process_group(self.dev, force=args.force_dp)     # RCCL; keeps its banner off stdout
            # gradient exchange: in-kernel peer-to-peer (hipIpc over xGMI) if it passes its self-check on
            # this node, RCCL all-reduce between the phases otherwise (NDP_DP_EXCHANGE=rccl|p2p forces one)
            if self.world > 1:
                self.p2p, self.reduce_fn, self.exchange = dp.make_exchange(self.dev, self.world, log=self.log)
            else:
                self.reduce_fn, self.exchange = dp.sum_all_reduce(), "rccl"
        from oracle import gan_oracle as O   # inputs + the step-0 parity figure + cpu_baseline only
        self.O = O
        self.g0, self.d0 = O.init_params(0, NZ)

    def log(self, msg):
        if self.rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def copy_stream(self):
        """ONE upload stream for every workload of this process that overlaps host-to-device copies with compute: the GPU
        runs a handful of hardware queues side by side and time-slices beyond that -- with the graph-upload stream, a second
        copy stream and the forward model's weight-gradient stream all in use, every kernel of the forward model ran 2-4x
        slower (4.47 instead of 1.90 ms per iteration; GPU_MAX_HW_QUEUES=2 restored it)."""
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(self.dev)
        return self._copy_stream

    def make_trainer(self, batch, k, global_flat, spl, p2p, reduce_fn, use_graph=True):
        from ndivplanning_amd.models.gan import Decoder, Discriminator
        from ndivplanning_amd.trainer import GanTrainer
        dec, dis = Decoder(NZ), Discriminator()
        dec.load_state_dict(self.g0)
        dis.load_state_dict(self.d0)
        dec, dis = dec.to(self.dev), dis.to(self.dev)
        if p2p is not None:
            p2p.reset()                # a new trainer counts its exchanges from 1 again: flags must be zero
        return GanTrainer(dec, dis, flat=batch * (TRAJ - 1), num_sample=k, flat_global=global_flat, reduce_fn=reduce_fn,
                          p2p=p2p, use_graph=use_graph, noise_seed=self.rank, steps_per_launch=spl,
                          copy_stream=self.copy_stream() if self.world == 1 else None)

    def fill_slots(self, tr, batch, k, seed0=2000):
        for slot in range(tr.nslots):
            c_, a_, _ = self.O.synthetic_batch(seed0 + 17 * slot + self.rank, batch, k, NZ, steps=1)
            tr.codes_slots[slot].copy_(c_)
            tr.actions_slots[slot].copy_(a_)

    @staticmethod
    def run_steps(tr, n, stepper=None):
        spl, done = tr.nslots, 0
        many = stepper or tr.step_many
        while spl > 1 and n - done >= spl:
            many()                     # spl iterations, one graph replay
            done += spl
        if stepper is None and spl > 1 and n - done > 1 and tr.use_graph and tr.reduce_fn is None:
            tr.step_many(count=n - done)   # the remainder as one replay too (a 20-step run = 16 + 4)
            done = n
        for _ in range(n - done):
            tr.step()                  # device noise, resident inputs

    def timed(self, tr, steps, warmup, stepper=None):
        """Warm up, then R x (barrier, EXACTLY `steps` steps, barrier), maximum over ranks per repetition;
        returns (median seconds per repetition, [seconds per repetition])."""
        self.barrier()
        self.run_steps(tr, max(warmup, 1), stepper)
        self.run_steps(tr, tr.nslots + 1, stepper)                        # both graphs exist before timing
        self.barrier()
        reps, total = [], 1
        want = self.args.min_timed_ms * 1e-3
        while True:
            self.barrier()
            t0 = time.perf_counter()
            self.run_steps(tr, steps, stepper)
            self.barrier()
            reps.append(self.max_over_ranks(time.perf_counter() - t0))
            if len(reps) == 1 and reps[0] < want:
                # every rank computes the same count from the max-reduced first repetition; odd, so that the
                # median is a measured repetition
                total = min(25, max(3, int(math.ceil(want / max(reps[0], 1e-6))))) | 1
            if len(reps) >= total:
                break
        return statistics.median(reps), reps


def step_flops(m):
    """FLOPs of the reference's step on m rows, whatever this implementation executes."""
    return 2.0 * MACS_PER_ROW_STEP * m


def step_flops_executed(m, k):
    return 2.0 * kernel_macs(k)[1] * m


def workload_summary(batch, k, world, sec, steps, reps, extra=None):
    m = batch * (TRAJ - 1) * k
    iters = steps / sec
    out = {"batch_per_gpu": batch, "num_sample": k, "rows_per_gpu": m, "global_batch": batch * world,
           "ms_per_step": round(1e3 * sec / steps, 5), "iterations_per_sec": round(iters, 2),
           "trajectories_per_sec": round(iters * batch * world, 1), "repeats": len(reps),
           "whole_step_tflops_per_gpu": round(step_flops(m) * iters / 1e12, 3),
           # the reference's step (629,760 MAC per row) per second against the fp32-MFMA peak ...
           "whole_step_frac_of_fp32_mfma_peak": round(step_flops(m) * iters / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
           # ... and the MACs the kernels really execute (real pass of D on 1/K of the rows): matrix-pipe utilisation
           "executed_frac_of_fp32_mfma_peak": round(step_flops_executed(m, k) * iters / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
    if extra:
        out.update(extra)
    return out


def kernel_table(b, tr, m, k, reps=50):
    """Per-kernel durations: HIP events around every launch, eager launches of the same step
    (every rank steps -- the exchange needs all of them -- rank 0 records)."""
    from ndivplanning_amd import _capi
    kernels = {}
    saved = tr.use_graph
    tr.use_graph = False
    if b.rank == 0:
        _capi.timing_enable(True)
    for _ in range(reps):
        tr.step()
    torch.cuda.synchronize(b.dev)
    if b.rank == 0:
        timed = _capi.timing_collect()
        _capi.timing_enable(False)
        table = kernel_macs(k)[0]
        for name, (ms, cnt) in timed.items():
            us = 1e3 * ms / max(cnt, 1)
            macs = table.get(name)
            kernels[name] = {"avg_us": round(us, 3), "launches_per_step": cnt / reps,
                             "tflops": round(2.0 * macs * m / (us * 1e-6) / 1e12, 3) if macs else None}
    tr.use_graph = saved
    return kernels


def large_m_point(b, batch, k, steps):
    """A large-M single-GPU workload with its per-kernel table (whole-step fraction of the MFMA peak)."""
    tr = b.make_trainer(batch, k, batch * (TRAJ - 1), 4, None, None)
    b.fill_slots(tr, batch, k, seed0=5000)
    sec, reps = b.timed(tr, steps, max(steps // 4, 8))
    m = batch * (TRAJ - 1) * k
    out = workload_summary(batch, k, 1, sec, steps, reps)
    out["kernels"] = kernel_table(b, tr, m, k, reps=10)
    del tr
    torch.cuda.empty_cache()
    return out


def config4_point(b, steps):
    """BASELINE configs[3]: image-conditioned generator, batch = 128, one GPU.  The reference encodes the current and the
    target frame of every FLAT row (train_gan.py:152-155: 2 x 896 images, the target S-fold redundantly); here the
    B x 8 = 1,024 unique frames go through the frozen encoder once (ndp_encoder_forward), then the fused step runs at
    M = 5,376.  Both parts are timed on the launch stream; their sum is the image-mode step."""
    from ndivplanning_amd.models.image_autoencoder import Encoder
    from ndivplanning_amd.train_gan import encode_batch
    batch, k = 128, 6
    torch.manual_seed(0)
    enc = Encoder().to(b.dev).eval()
    gen = torch.Generator(device="cpu").manual_seed(4)
    frames = (torch.rand(batch, TRAJ, 3, 128, 128, generator=gen) * 2.0 - 1.0).to(b.dev)
    n_img = batch * TRAJ
    flat_imgs = frames.reshape(n_img, 3, 128, 128)
    with torch.no_grad():
        codes128 = enc(flat_imgs)                                         # warm-up (packs parameters, allocates)
        torch.cuda.synchronize(b.dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        nrep = 5
        ev[0].record()
        for _ in range(nrep):
            codes128 = enc(flat_imgs)
        ev[1].record()
        torch.cuda.synchronize(b.dev)
    enc_ms = ev[0].elapsed_time(ev[1]) / nrep
    codes = encode_batch(codes128.reshape(batch, TRAJ, 128), None, TRAJ)  # [896, 256]
    assert tuple(codes.shape) == (batch * (TRAJ - 1), 256) and bool(torch.isfinite(codes).all())
    tr = b.make_trainer(batch, k, batch * (TRAJ - 1), 4, None, None)
    gen_a = torch.Generator().manual_seed(5)
    actions = torch.rand(batch * (TRAJ - 1), 4, generator=gen_a) * 2.0 - 1.0
    for slot in range(tr.nslots):
        tr.codes_slots[slot].copy_(codes)
        tr.actions_slots[slot].copy_(actions)
    sec, reps = b.timed(tr, steps, max(steps // 4, 8))
    step_ms = 1e3 * sec / steps
    losses = tr.losses()
    enc_tflops = ENCODER_FLOP_PER_IMAGE * n_img / (enc_ms * 1e-3) / 1e12
    m = batch * (TRAJ - 1) * k

    # ---- the same iteration END TO END, upload included (the reference uploads every batch: train_gan.py:119-124).
    # Frames leave the loader as decoded bytes [B,8,128,128,3]; they are uploaded as bytes (a quarter of the reference's
    # float upload) into one of two device buffers by a copy stream while the other buffer's batch is encoded and trained
    # on; the first convolution normalises them as it gathers (ndp_encoder_forward_u8).
    gen8 = torch.Generator().manual_seed(6)
    pool = [torch.randint(0, 256, (n_img, 128, 128, 3), generator=gen8, dtype=torch.uint8).pin_memory() for _ in range(3)]
    dev_buf = [torch.empty(n_img, 128, 128, 3, dtype=torch.uint8, device=b.dev) for _ in range(2)]
    host_f32 = torch.empty(n_img, 3, 128, 128).pin_memory()
    dev_f32 = torch.empty(n_img, 3, 128, 128, device=b.dev)
    copy_stream = b.copy_stream()
    main = torch.cuda.current_stream(b.dev)

    def time_copy(dst, src, reps=5):
        torch.cuda.synchronize(b.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize(b.dev)
        return e0.elapsed_time(e1) / reps
    h2d_u8_ms, h2d_f32_ms = time_copy(dev_buf[0], pool[0]), time_copy(dev_f32, host_f32)
    del host_f32, dev_f32
    uploaded = [torch.cuda.Event(), torch.cuda.Event()]
    consumed = [torch.cuda.Event(), torch.cuda.Event()]

    def upload(i):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[i % 2])                    # the batch that last used this buffer is done with it
            dev_buf[i % 2].copy_(pool[i % 3], non_blocking=True)
            uploaded[i % 2].record(copy_stream)

    def iteration(i):
        upload(i + 1)                                                 # next batch's upload runs beside this batch's work
        main.wait_event(uploaded[i % 2])
        with torch.no_grad():
            c128 = enc(dev_buf[i % 2])
        consumed[i % 2].record(main)
        tr.step(encode_batch(c128.reshape(batch, TRAJ, 128), None, TRAJ), actions_dev)
    actions_dev = actions.to(b.dev)
    consumed[0].record(main)
    consumed[1].record(main)
    upload(0)
    for i in range(3):
        iteration(i)
    torch.cuda.synchronize(b.dev)
    n_it = 12
    t0 = time.perf_counter()
    for i in range(3, 3 + n_it):
        iteration(i)
    torch.cuda.synchronize(b.dev)
    e2e_ms = 1e3 * (time.perf_counter() - t0) / n_it
    out = {"workload": "BASELINE configs[3]: image-conditioned step, batch=128 trajectories x 8 frames of 3x128x128, "
                       "num_sample=6; synthetic images, seeded random-init encoder",
           "unique_images": n_img, "rows": m,
           "encoder_ms": round(enc_ms, 4), "encoder_tflops": round(enc_tflops, 2),
           "encoder_frac_of_fp32_mfma_peak": round(enc_tflops / MFMA_F32_PEAK_TFLOPS, 4),
           "step_ms": round(step_ms, 5),
           "step_frac_of_fp32_mfma_peak": round(step_flops(m) / (step_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
           "image_step_ms": round(enc_ms + step_ms, 4),
           "image_steps_per_sec": round(1e3 / (enc_ms + step_ms), 2),
           "h2d_ms": round(h2d_u8_ms, 4),
           "h2d_note": "upload of one batch's %d frames from pinned host memory as decoded bytes [n,128,128,3] (%.1f MB); as "
                       "the reference's float tensors (%.1f MB) it takes %.3f ms" % (n_img, n_img * 49152 / 1e6,
                                                                                     n_img * 196608 / 1e6, h2d_f32_ms),
           "image_step_ms_with_upload": round(e2e_ms, 4),
           "image_steps_per_sec_with_upload": round(1e3 / e2e_ms, 2),
           "with_upload_how": "every iteration uploads a fresh batch of byte frames (double-buffered, copy stream), encodes "
                              "it (conv1 normalises while gathering) and runs the fused step: wall clock per iteration",
           "cached_codes_steps_per_sec": round(1e3 / step_ms, 2),
           "flops_as_reference_writes_it": "2 x 896 encoder passes = %.3f TFLOP/step; de-duplicated %.3f TFLOP/step"
                                           % (ENCODER_FLOP_PER_IMAGE * 2 * 896 / 1e12, ENCODER_FLOP_PER_IMAGE * n_img / 1e12),
           "last_losses": {"D": losses[0], "G": losses[1], "ndiv": losses[2]}}
    del tr, enc, frames, pool, dev_buf
    torch.cuda.empty_cache()
    return out


# forward (next-frame) model, SURVEY.md section 8 row f4: MACs per image of one forward pass, layer by layer
# (models/forward_encoder.py:20-97): conv1..6, deconv1..6, conv_refine_1, conv_refine_2
FM_FWD_MACS = (64 * 64 * 64 * 27 + 4 * 75_497_472 + 128 * 16384            # encoder: 311.2 M
               + 132 * 1024 * 16 + 5 * 268_435_456                          # deconv1, deconv2..6
               + 128 * 128 * 16 * 288 + 128 * 128 * 3 * 144)                # refinement
FM_STEP_FLOP_PER_IMAGE = 2.0 * (3 * FM_FWD_MACS - 64 * 64 * 64 * 27)        # + data and weight gradients (conv1 has no data gradient)


def forward_model_point(b, cpu_seconds):
    """One iteration of train_forward_model.py:98-112 (forward, MSE, backward, Adam) at the reference's batch of 8 images
    and at 32: ForwardModelTrainer.step on resident synthetic frames, timed with events on the launch stream; the kernel
    table comes from the library's per-launch events (ndp_timing_*).  FLOPs are the reference's (2 x MACs of the
    convolutions / transposed convolutions, forward + both gradients)."""
    from ndivplanning_amd import _capi
    from ndivplanning_amd.forward_trainer import ForwardModelTrainer
    from ndivplanning_amd.models import forward_encoder as FE
    out = {"workload": "train_forward_model.py:98-112, one frame pair: ForwardAutoencoder forward (training-mode BatchNorm), "
                       "MSE, backward, Adam; synthetic 3x128x128 frames, weight_init(0, 0.02)",
           "gflop_per_image": round(FM_STEP_FLOP_PER_IMAGE / 1e9, 3)}
    for n, steps in ((8, 40), (32, 16)):
        torch.manual_seed(0)
        model = FE.ForwardAutoencoder()
        model.decoder.weight_init(0.0, 0.02)
        model.encoder.weight_init(0.0, 0.02)
        tr = ForwardModelTrainer(model.to(b.dev).train(), batch=n)
        gen = torch.Generator().manual_seed(1)
        cur, fut = ((torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1).to(b.dev) for _ in range(2))
        act = (torch.rand(n, 4, generator=gen) * 2 - 1).to(b.dev)
        for _ in range(3):
            tr.step(cur, fut, act)
        torch.cuda.synchronize(b.dev)
        reps = []
        for _ in range(5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(steps):
                tr.step(cur, fut, act)
            ev[1].record()
            torch.cuda.synchronize(b.dev)
            reps.append(ev[0].elapsed_time(ev[1]) / steps)
        ms = statistics.median(reps)
        was = _capi.load().ndp_fm_side_stream(0)               # one stream: the per-launch durations do not overlap
        _capi.timing_enable(True)
        for _ in range(3):
            tr.step(cur, fut, act)
        torch.cuda.synchronize(b.dev)
        timed = _capi.timing_collect()
        _capi.timing_enable(False)
        _capi.load().ndp_fm_side_stream(was)
        groups = {}
        for name, (tot, cnt) in timed.items():
            key = name.split("[")[0]
            g = groups.setdefault(key, [0.0, 0])
            g[0] += tot
            g[1] += cnt
        total = sum(v[0] for v in groups.values())
        tflops = FM_STEP_FLOP_PER_IMAGE * n / (ms * 1e-3) / 1e12
        heavy = {nm: v for nm, v in timed.items() if nm.startswith(("k_fm_gemm[deconv", "k_fm_dgrad[deconv", "k_fm_wgrad[deconv"))
                 and nm[-2] in "2345"}
        heavy_tflops = {nm: round(2.0 * 268_435_456 * n / (v[0] / v[1] * 1e-3) / 1e12, 1) for nm, v in heavy.items()}
        point = {"batch": n, "ms_per_step": round(ms, 4), "steps_per_sec": round(1e3 / ms, 2),
                 "images_per_sec": round(n * 1e3 / ms, 1), "tflops": round(tflops, 2),
                 "frac_of_fp32_mfma_peak": round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
                 "repeat_ms_per_step": [round(r, 4) for r in reps], "loss": float(tr.loss.item()),
                 "launches_per_step": sum(v[1] for v in groups.values()) // 3,
                 "kernel_time_share": {k_: {"us_per_step": round(v[0] / 3 * 1e3, 1), "launches_per_step": v[1] // 3,
                                            "share": round(v[0] / total, 4)}
                                       for k_, v in sorted(groups.items(), key=lambda kv: -kv[1][0])},
                 "deconv2to5_tflops": {"min": min(heavy_tflops.values()), "max": max(heavy_tflops.values()),
                                       "note": "the 12 launches of 268 M MAC per image each (deconv2..5 forward, data and "
                                               "weight gradient), from their HIP-event durations; fp32 MFMA peak %.1f"
                                               % MFMA_F32_PEAK_TFLOPS}}
        out["batch%d" % n] = point
        del tr, model
        torch.cuda.empty_cache()
    if cpu_seconds > 0:
        from oracle import forward_model_oracle as FO
        n = 8
        state = FO.init_forward_model_state(0)
        oracle = FO.ForwardModelTrainer(state, lr=2e-4)
        gen = torch.Generator().manual_seed(1)
        cur, fut = (torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1 for _ in range(2))
        act = torch.rand(n, 4, generator=gen) * 2 - 1
        oracle.step(cur, fut, act)
        cnt, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < cpu_seconds:
            oracle.step(cur, fut, act)
            cnt += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(cnt / dt, 3), "unit": "steps/s", "kind": "port", "threads": torch.get_num_threads(),
                               "sample": "%d iterations at batch 8 of the oracle's restatement (torch %s CPU fp32 conv / "
                                         "conv_transpose / batch_norm + Adam) in %.1f s" % (cnt, torch.__version__, dt),
                               "gpu_over_cpu": round(out["batch8"]["steps_per_sec"] / (cnt / dt), 1)}
    return out


def h2d_point(b, batch, k, steps, spl):
    """Config 2 with a fresh batch per step uploaded from pinned host memory (what the reference's loop does per
    iteration, train_gan.py:119-124), the upload of launch i+1 overlapped with the graph of launch i."""
    tr = b.make_trainer(batch, k, batch * (TRAJ - 1), spl, None, None)
    flat = batch * (TRAJ - 1)
    gen = torch.Generator().manual_seed(77)
    pool = 4                                                              # rotating host batches of spl steps each
    host_c = [torch.randn(spl, flat, 256, generator=gen).pin_memory() for _ in range(pool)]
    host_a = [(torch.rand(spl, flat, 4, generator=gen) * 2.0 - 1.0).pin_memory() for _ in range(pool)]
    b.fill_slots(tr, batch, k)
    state = {"i": 0}

    def stepper():
        i = state["i"] = (state["i"] + 1) % pool
        tr.step_many_from_host(host_c[i], host_a[i])
    n = max(steps - steps % spl, spl)
    sec, reps = b.timed(tr, n, max(steps // 10, spl), stepper)
    out = {"ms_per_step": round(1e3 * sec / n, 5), "steps_per_sec": round(n / sec, 2), "repeats": len(reps),
           "bytes_per_step": flat * (256 + 4) * 4, "steps_per_upload": spl,
           "how": "two alternating sets of input slots, each with its own captured graph: the copy stream uploads the "
                  "next launch's batches from pinned host memory into the idle set while the other set's graph runs "
                  "(GanTrainer.step_many_from_host)"}
    del tr
    torch.cuda.empty_cache()
    return out


def child_rank_env(base_env, rank, world, port):
    """Environment of child rank `rank` of a self-launched N-rank run: the variables torch.distributed.run would set
    (one process per GPU, rendezvous on the loopback address -- the container hostname may not resolve)."""
    env = dict(base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "NDP_BENCH_CHILD": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": base_env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    env.setdefault("OMP_NUM_THREADS", "8")
    return env


def child_rank_argv(argv):
    """Command line of one child rank: this interpreter, this file, the parent's own arguments unchanged."""
    return [sys.executable, os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a torch.distributed.run environment: this process never touches a GPU; it
    starts N fresh rank processes (never an exec of a process that has initialised the runtime), relays rank 0's one
    JSON line on stdout, forwards every rank's stderr, enforces a wall-clock limit, and exits non-zero with the
    failing rank's stderr tail when a rank fails."""
    import socket
    import subprocess
    import tempfile
    import threading
    world = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    limit = float(os.environ.get("NDP_BENCH_LAUNCH_TIMEOUT_S", "1500"))
    tails = [[] for _ in range(world)]
    procs, threads = [], []
    rank0_out = tempfile.TemporaryFile(mode="w+")

    def pump(rank, stream):
        for line in stream:
            tails[rank].append(line)
            del tails[rank][:-40]
            sys.stderr.write(line if rank == 0 else "[rank %d] %s" % (rank, line))
            sys.stderr.flush()

    for rank in range(world):
        p = subprocess.Popen(child_rank_argv(argv), env=child_rank_env(os.environ, rank, world, port), cwd=ROOT,
                             stdin=subprocess.DEVNULL, stdout=rank0_out if rank == 0 else subprocess.DEVNULL,
                             stderr=subprocess.PIPE, text=True, start_new_session=True)
        procs.append(p)
        t = threading.Thread(target=pump, args=(rank, p.stderr), daemon=True)
        t.start()
        threads.append(t)

    def stop_all():
        for p in procs:                      # exactly the processes started above (each its own session / group)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 15)
                except OSError:
                    pass
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass

    t0, failed = time.time(), None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = "rank %d exited with code %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > limit:
                failed = "wall-clock limit of %.0f s reached" % limit
                break
            time.sleep(0.05)
    finally:
        if failed:
            stop_all()
    for t in threads:
        t.join(timeout=5)
    rank0_out.seek(0)
    lines = [ln for ln in rank0_out.read().splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    if failed is None and not json_lines:
        failed = "rank 0 printed no JSON line"
    if failed:
        r = next((r for r, p in enumerate(procs) if p.returncode not in (None, 0)), 0)
        sys.stderr.write("[bench] %d-rank run failed: %s\n---- stderr tail of rank %d ----\n%s" % (world, failed, r, "".join(tails[r])))
        sys.exit(1)
    print(json_lines[-1], flush=True)
    sys.exit(0)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])          # never returns; no GPU call has happened in this process
    b = Bench(args)
    world, rank, dev, dist, O = b.world, b.rank, b.dev, b.dist, b.O
    batch, k = args.batch, args.num_sample
    flat = batch * (TRAJ - 1)
    m = flat * k
    graph_ok = not args.no_graph
    spl = args.steps_per_launch if (b.reduce_fn is None and graph_ok) else 1
    if spl > 1 and args.steps % spl != 0 and args.steps <= 3 * spl:
        spl = args.steps          # a short bracket that is not a multiple of the default: one graph of exactly K steps
    tr = b.make_trainer(batch, k, flat * world, spl, b.p2p, b.reduce_fn, use_graph=graph_ok)

    # step-0 parity figure (outside the timed region): NDiv / losses vs the oracle on rank 0's shard
    parity = None
    codes, actions, noise = O.synthetic_batch(1000 + rank, batch, k, NZ, steps=1)
    if world == 1:
        ref = O.StepMath({n_: v.clone() for n_, v in b.g0.items()}, {n_: v.clone() for n_, v in b.d0.items()})
        out = ref.step(codes, actions, noise[0])
        tr.step(codes.to(dev), actions.to(dev), noise[0].to(dev))
        dl, gl, pd = tr.losses()
        parity = {"ndiv_rel_err": abs(pd - out["pair_div"].item()) / max(1.0, abs(out["pair_div"].item())),
                  "d_loss_abs_err": abs(dl - out["d_loss"].item()), "g_loss_abs_err": abs(gl - out["g_loss"].item()),
                  "action_hat_max_abs_err": (tr.action_hat[:m].cpu() - out["action_hat"]).abs().max().item()}
        del tr
        tr = b.make_trainer(batch, k, flat * world, spl, b.p2p, b.reduce_fn, use_graph=graph_ok)
    tr.codes.copy_(codes)
    tr.actions.copy_(actions)
    b.fill_slots(tr, batch, k)

    if b.p2p is not None:
        # a wait that timed out during warm-up (status word) means the exchange does not work on this
        # node although its self-check passed: fall back to the RCCL path rather than time garbage
        b.barrier()
        b.run_steps(tr, max(args.warmup, spl))
        bad = torch.tensor([b.p2p.status()], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) != 0:
            b.log("peer-to-peer exchange timed out in warm-up (%s); falling back to RCCL" % (b.p2p.diagnostics(),))
            del tr
            b.p2p.close()
            b.p2p, b.reduce_fn, b.exchange, spl = None, b.dp.sum_all_reduce(), "rccl (p2p timed out in warm-up)", 1
            tr = b.make_trainer(batch, k, flat * world, spl, None, b.reduce_fn, use_graph=graph_ok)
            b.fill_slots(tr, batch, k)
    def replicas_in_lockstep(t_):
        # the replicas must have stayed bit-identical: compare checksums of the parameter bits
        return b.dp.replicas_bit_identical([t_.g_flat, t_.d_flat])

    p2p_note = None
    while True:
        elapsed, reps = b.timed(tr, args.steps, args.warmup)
        losses = tr.losses()
        hip_graph = bool(tr.use_graph)
        kernels = kernel_table(b, tr, m, k)
        replicas_identical = replicas_in_lockstep(tr) if world > 1 else None
        if b.p2p is None:
            break
        # the in-kernel exchange has never met this node before today: a timed-out wait or diverged replicas void
        # the measurement just taken -- say so and take it again through the collective rather than report it
        bad = torch.tensor([1 if (b.p2p.status() != 0 or not replicas_identical) else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) == 0:
            break
        p2p_note = "p2p failed during the timed run (status %d, replicas identical %s, %s): re-timed through rccl" % (
            b.p2p.status(), replicas_identical, b.p2p.diagnostics())
        b.log(p2p_note)
        del tr
        b.p2p.close()
        b.p2p, b.reduce_fn, b.exchange, spl = None, b.dp.sum_all_reduce(), "rccl (p2p failed in the timed run)", 1
        tr = b.make_trainer(batch, k, flat * world, spl, None, b.reduce_fn, use_graph=graph_ok)
        b.fill_slots(tr, batch, k)
    del tr
    torch.cuda.empty_cache()

    # ---- further workloads (every rank takes part in the N > 1 ones; failures never cost the headline line)
    extras, extra_errors = {}, {}

    only = [x for x in os.environ.get("NDP_BENCH_EXTRAS", "").split(",") if x]     # diagnostics: run just these extras

    def extra(name, fn):
        if only and name not in only:
            return
        try:
            extras[name] = fn()
        except Exception as exc:                                           # noqa: BLE001
            extra_errors[name] = repr(exc)[:300]
            b.log("extra %s failed: %r" % (name, exc))
    if not args.no_extras:
        if world > 1:
            nsteps = min(args.steps, 400)

            def dp_point(pb, pk, global_batch):
                spl_ = args.steps_per_launch if (b.reduce_fn is None and graph_ok) else 1
                t_ = b.make_trainer(pb, pk, global_batch * (TRAJ - 1), spl_, b.p2p, b.reduce_fn, use_graph=graph_ok)
                b.fill_slots(t_, pb, pk, seed0=7000)
                sec_, reps_ = b.timed(t_, nsteps, max(nsteps // 4, spl_))
                ok_ = replicas_in_lockstep(t_)
                if b.p2p is not None:
                    b.p2p.check()
                out_ = workload_summary(pb, pk, world, sec_, nsteps, reps_,
                                        {"gradient_exchange": b.exchange, "replicas_bit_identical": ok_})
                out_["global_steps_per_sec"] = out_["iterations_per_sec"]
                del t_
                torch.cuda.empty_cache()
                return out_
            if 256 % world == 0:
                # BASELINE configs[2]: STRONG scaling -- the global batch of 256 split over the ranks (32 per GPU at N = 8)
                extra("strong_config3", lambda: dict(dp_point(256 // world, 6, 256), scaling="strong",
                                                     workload="BASELINE configs[2]: global batch 256 over %d ranks" % world))
            # BASELINE configs[4]: K = 32, 128 trajectories per GPU (global 1,024 at N = 8)
            extra("config5_shard", lambda: dict(dp_point(128, 32, 128 * world), scaling="weak",
                                                workload="BASELINE configs[4] per-GPU shard: batch 128 x K=32 per rank"))
            if b.p2p is not None:
                # the same headline workload through the collective instead (8 eager kernels + 2 all-reduces per step)
                def rccl_headline():
                    t_ = b.make_trainer(batch, k, flat * world, 1, None, b.dp.sum_all_reduce(), use_graph=graph_ok)
                    b.fill_slots(t_, batch, k)
                    sec_, reps_ = b.timed(t_, nsteps, max(nsteps // 4, 8))
                    out_ = workload_summary(batch, k, world, sec_, nsteps, reps_, {"gradient_exchange": "rccl"})
                    out_["value_steps_per_sec"] = round(out_["iterations_per_sec"] * world, 2)
                    del t_
                    return out_
                extra("headline_through_rccl", rccl_headline)

            # the forward (next-frame) model trained data-parallel, 8 images per rank: the 133 MB flat gradient averaged
            # (a) bucket by bucket on a communication stream while the backward pass is still running (the default of the
            # train_forward_model.py mirror), (b) by ONE all-reduce between backward and Adam; and the same iteration with
            # no exchange at all, so that the part of the all-reduce that is NOT hidden can be read off
            def forward_model_dp():
                from ndivplanning_amd.forward_trainer import ForwardModelTrainer
                from ndivplanning_amd.models import forward_encoder as FE
                n_, steps_ = 8, 20
                gen = torch.Generator().manual_seed(100 + rank)
                cur, fut = ((torch.rand(n_, 3, 128, 128, generator=gen) * 2 - 1).to(dev) for _ in range(2))
                act = (torch.rand(n_, 4, generator=gen) * 2 - 1).to(dev)

                def run(kind):
                    torch.manual_seed(0)
                    model = FE.ForwardAutoencoder()
                    model.decoder.weight_init(0.0, 0.02)
                    model.encoder.weight_init(0.0, 0.02)
                    kw = {"single": dict(reduce_fn=b.dp.mean_all_reduce(world)),
                          "bucketed": dict(bucket_reduce=b.dp.BucketedMeanAllReduce(world)),
                          "bucketed_cross_rank_batchnorm": dict(bucket_reduce=b.dp.BucketedMeanAllReduce(world),
                                                                sync_batchnorm_world=world), "none": {}}[kind]
                    t_ = ForwardModelTrainer(model.to(dev).train(), batch=n_, **kw)
                    for _ in range(3):
                        t_.step(cur, fut, act)
                    b.barrier()
                    t0 = time.perf_counter()
                    for _ in range(steps_):
                        t_.step(cur, fut, act)
                    b.barrier()
                    sec_ = b.max_over_ranks(time.perf_counter() - t0)
                    same = b.dp.replicas_bit_identical([t_.params]) if kind != "none" else None
                    mb = t_.grad.numel() * 4 / 1e6
                    t_.close()
                    del t_, model
                    torch.cuda.empty_cache()
                    return 1e3 * sec_ / steps_, same, mb
                base_ms, _, mb = run("none")
                out_ = {"workload": "train_forward_model.py iteration, 8 images per rank, mean all-reduce of the flat gradient "
                                    "(%.0f MB)" % mb, "scaling": "weak", "ms_per_step_without_exchange": round(base_ms, 4)}
                for kind in ("bucketed", "single", "bucketed_cross_rank_batchnorm"):
                    ms_, same, _ = run(kind)
                    out_[kind] = {"ms_per_step": round(ms_, 4), "global_images_per_sec": round(n_ * world * 1e3 / ms_, 1),
                                  "exposed_all_reduce_ms": round(ms_ - base_ms, 4), "replicas_bit_identical": same}
                out_["replicas_bit_identical"] = bool(out_["bucketed"]["replicas_bit_identical"] and out_["single"]["replicas_bit_identical"])
                out_["how"] = ("bucketed: 7 ranges of the flat gradient, each all-reduced on a communication stream as soon as the "
                               "backward pass has completed it (ndp_fm_grad_buckets / ndp_fm_bucket_wait); single: one "
                               "collective between backward and Adam; bucketed_cross_rank_batchnorm: plus the BatchNorm statistics "
                               "summed over the ranks (20 int64 all-reduces per iteration, ndp_fm_set_stat_sync): the ranks train "
                               "the single-process step on the global batch -- train_forward_model.py's default")
                return out_
            extra("forward_model_dp", forward_model_dp)
        else:
            # 64 steps per upload: every launch costs ~250-340 us beyond its graph (upload not fully hidden, event wait
            # ahead of the replay), whatever its size -- measured 10,630 / 10,770 / 11,940 steps/s at 16 / 32 / 64
            extra("h2d_per_launch", lambda: h2d_point(b, batch, k, max(min(args.steps, 2560), 640), 64))
            extra("config4", lambda: config4_point(b, 200))
            extra("large_m", lambda: {"config5_shard_b128_k32": large_m_point(b, 128, 32, 60),
                                      "b1024_k6": large_m_point(b, 1024, 6, 60)})
            extra("forward_model", lambda: forward_model_point(b, 0.0 if args.no_cpu_baseline else min(args.cpu_seconds, 8.0)))
    if world > 1 and b.p2p is not None:
        b.p2p.close()
    if rank != 0:
        dist.destroy_process_group()
        return

    iters_per_s = args.steps / elapsed
    value = iters_per_s * world
    table = kernel_macs(k)[0]
    dom = max((n_ for n_ in kernels if table.get(n_)), key=lambda n_: kernels[n_]["avg_us"] * kernels[n_]["launches_per_step"])
    dom_flops = 2.0 * table[dom] * m                     # what the kernel executes per launch
    achieved = dom_flops / (kernels[dom]["avg_us"] * 1e-6) / 1e12
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
                   "repeats": len(reps), "repeat_ms_per_step": [round(1e3 * r / args.steps, 5) for r in reps],
                   "reported": "median repetition of %d x exactly %d steps, each bracketed by barrier + synchronize"
                               % (len(reps), args.steps),
                   "gradient_exchange": b.exchange, "all_reduce_us": dict(b.dp.last_exchange_report) or None,
                   "replicas_bit_identical": replicas_identical, "exchange_note": p2p_note,
                   "last_losses": {"D": losses[0], "G": losses[1], "ndiv": losses[2]}},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": MFMA_F32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                     # the same launch priced at the reference's work (D's real pass on all M rows, as round 1 counted it
                     # before the real pass was deduplicated): comparable across rounds
                     "achieved_reference_equivalent": round(achieved * kernel_macs(1)[0][dom] / table[dom], 3),
                     "frac_reference_equivalent": round(achieved * kernel_macs(1)[0][dom] / table[dom] / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": PMC_HBM_BYTES_DEFAULT.get(dom) if (batch, k) == (64, 6) else None,
                     "traffic_source": PMC_SOURCE,
                     "mfma_busy_counter": PMC_MFMA_BUSY_DEFAULT.get(dom) if (batch, k) == (64, 6) else None,
                     "mfma_busy_source": PMC_MFMA_SOURCE,
                     "algorithmic_flops_per_launch": dom_flops,
                     "flops_note": "per-kernel figures count the MACs executed: D's real pass runs on the FLAT distinct "
                                   "rows (1/K of what the reference feeds through D, train_gan.py:140-156)",
                     "whole_step": {"flops": step_flops(m),
                                    "tflops": round(step_flops(m) * iters_per_s / 1e12, 3),
                                    "frac": round(step_flops(m) * iters_per_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                    "flops_executed": step_flops_executed(m, k),
                                    "frac_executed": round(step_flops_executed(m, k) * iters_per_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                     "kernels": kernels},
    }
    if parity is not None:
        result["parity_step0"] = {k_: float("%.3e" % v) for k_, v in parity.items()}
    result.update(extras)
    if extra_errors:
        result["extras_failed"] = extra_errors
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(batch, k, NZ, args.cpu_seconds)
        result["cpu_baseline"]["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
