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
            with _stdout_to_stderr():                       # gloo reports its connections on stdout too
                dist.init_process_group(backend=backend, **kw)
                dist.barrier()
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


def mean_all_reduce(world, group=None):
    """reduce_fn for ForwardModelTrainer: in-place MEAN all-reduce of a flat gradient -- every rank's loss is the mean over
    its own shard (train_forward_model.py:106-107), equal shards, so the mean of the ranks' gradients is the gradient of
    the global-batch mean (what torch's DistributedDataParallel does; BatchNorm statistics stay per rank, as there
    without SyncBatchNorm)."""
    def reduce_fn(flat_grad):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        flat_grad.mul_(1.0 / world)
    return reduce_fn


class BucketedMeanAllReduce:
    """Forward-model gradient exchange overlapped with the backward pass (ForwardModelTrainer(bucket_reduce=...)).

    The library records an event per gradient bucket where the backward pass completes it (ndp_fm_grad_buckets: 7 ranges
    of the flat vector, last layers first; include/ndp.h).  Called right after ndp_fm_train_grads has ENQUEUED the pass,
    this makes a communication stream wait for each bucket's event in turn and all-reduces (RCCL) and averages that range
    there, while the launch stream is still computing the earlier layers' gradients; the launch stream then waits for the
    communication stream, so what follows (Adam) sees the reduced vector.  Averaging a range elementwise is the same
    arithmetic whatever the bucketing: with two ranks (one addition per element) the result is bit-identical to
    mean_all_reduce's single collective; with more, RCCL's summation order per element may differ between message sizes,
    the replicas still agree with each other bit for bit.

    With a backend that stages CUDA tensors through the host (gloo, the one-GPU rehearsals) each collective blocks the
    host until its bucket is complete: correct, but not overlapped."""

    def __init__(self, world, group=None):
        self.world, self.group = int(world), group
        self.stream = None
        self.buckets = None

    def __call__(self, flat_grad, device):
        from . import _capi
        lib = _capi.load()
        if self.buckets is None:
            self.buckets = _capi.fm_grad_buckets()
            if sum(c for _, c in self.buckets) != flat_grad.numel():
                raise _capi.NdpError("gradient buckets do not cover the flat gradient")
        if self.stream is None:
            self.stream = torch.cuda.Stream(device)
        main = torch.cuda.current_stream(device)
        scale = 1.0 / self.world
        with torch.cuda.device(device), torch.cuda.stream(self.stream):
            for b, (off, cnt) in enumerate(self.buckets):
                _capi.check(lib.ndp_fm_bucket_wait(b, _capi.stream_ptr(device)), "ndp_fm_bucket_wait")
                piece = flat_grad[off:off + cnt]
                dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
                piece.mul_(scale)
        main.wait_stream(self.stream)


class CrossRankBatchNorm:
    """Forward-model BatchNorm statistics over ALL ranks' images (include/ndp.h, ndp_fm_set_stat_sync): the library calls
    back between the launch that fills a BatchNorm's fixed-point accumulator and the launch that reads it, and this
    all-reduces (SUM, int64: exact, order-free) the accumulator in place on the launch stream.  W ranks with B / W images
    each then train exactly the single-process step on B images (up to fp32 summation order inside a tile): what the
    reference's `batch_size` means.  20 small collectives per iteration, on the critical path.

    `workspace`: the trainer's workspace tensor (the accumulators live inside it).  Process-wide while installed;
    `close()` (or a new instance) removes it.  An exception inside a callback is kept and re-raised by `check()`."""

    def __init__(self, workspace, world, group=None):
        from . import _capi
        self._capi, self.lib = _capi, _capi.load()
        self.workspace, self.world, self.group = workspace, int(world), group
        self.error = None
        self.calls = 0

        def sync(acc, words, stream, ctx):
            try:
                off = (int(acc) - self.workspace.data_ptr()) // 4
                if off < 0 or off + 2 * int(words) > self.workspace.numel() or (int(acc) - self.workspace.data_ptr()) % 8:
                    raise RuntimeError("statistics accumulator outside the workspace")
                view = self.workspace[off:off + 2 * int(words)].view(torch.int64)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                self.calls += 1
            except Exception as exc:                           # noqa: BLE001 - must not propagate through the C frame
                if self.error is None:
                    self.error = exc
        self._cb = _capi.STAT_SYNC_FN(sync)                    # keep the thunk alive as long as it is installed
        import ctypes
        _capi.check(self.lib.ndp_fm_set_stat_sync(ctypes.cast(self._cb, ctypes.c_void_p), None, self.world),
                    "ndp_fm_set_stat_sync")

    def check(self):
        if self.error is not None:
            err, self.error = self.error, None
            raise RuntimeError("cross-rank BatchNorm statistics: %r" % (err,))

    def close(self):
        if self._cb is not None:
            self.lib.ndp_fm_set_stat_sync(None, None, 1)
            self._cb = None


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


def replicas_bit_identical(tensors, group=None):
    """True on EVERY rank iff the given (parameter) tensors hold the same BITS on all ranks: two int64 checksums of the
    bit patterns (plain sum, position-weighted sum), MIN- and MAX-reduced.  Data-parallel replicas apply the same Adam
    update to the same summed gradient, so they must stay bit-identical; a gradient exchange that delivered a stale or
    torn sum to one rank shows up here.  Collective (two small all-reduces); synchronises."""
    bits = torch.cat([t.detach().reshape(-1) for t in tensors]).contiguous().view(torch.int32).to(torch.int64)
    weights = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
    mine = torch.stack([bits.sum(), (bits * weights).sum()])
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return True
    if dist.get_backend(group) != "nccl":
        mine = mine.cpu()
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


class ReplicaDivergence(RuntimeError):
    """Data-parallel replicas no longer hold identical parameters: the gradient exchange named in the message
    delivered different sums to different ranks."""


def assert_replicas_identical(tensors, exchange, where, group=None):
    """Raise ReplicaDivergence on every rank (the comparison is collective, so all ranks see the same verdict)."""
    if not replicas_bit_identical(tensors, group):
        raise ReplicaDivergence(
            "data-parallel replicas diverged (%s): the '%s' gradient exchange gave the ranks different sums; rerun with "
            "NDP_DP_EXCHANGE=rccl to use the collective" % (where, exchange))


class P2PExchange:
    """The in-kernel gradient exchange of include/ndp.h ("peer-to-peer gradient exchange"): one
    uncached region per rank, mapped into every peer with hipIpc; the step's slab-reduce kernels
    then sum the gradients over ranks themselves, so a data-parallel step has the launches of a
    single-GPU step and replays as one HIP graph.  One process per GPU of ONE node (xGMI or, in
    tests, several processes sharing a GPU).  Needs an initialised torch.distributed group for
    the handle exchange only.  `GanTrainer(p2p=...)` takes the object."""

    def __init__(self, device, group=None, timeout_ms=10000):
        from . import _capi
        self._capi = _capi
        self.lib = _capi.load()
        self.device = torch.device(device)
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if not 1 <= self.world <= _capi.P2P_MAX_RANKS:
            raise ValueError("peer-to-peer exchange supports up to %d ranks, got %d" % (_capi.P2P_MAX_RANKS, self.world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import ctypes
        self._mapped = []
        self._region = ctypes.c_void_p()
        self._status_host = None
        self.shared_device, self.bus_ids = False, []
        # Every rank runs the SAME sequence of collectives whatever fails locally (a rank that raised before a
        # collective would leave the others waiting in it): local errors are carried through and raised together.
        err = None
        handle = ctypes.create_string_buffer(_capi.P2P_HANDLE_BYTES)
        with torch.cuda.device(self.device):
            try:
                _capi.check(self.lib.ndp_p2p_region_alloc(ctypes.byref(self._region)), "ndp_p2p_region_alloc")
                _capi.check(self.lib.ndp_p2p_export(self._region, handle), "ndp_p2p_export")
            except Exception as exc:                                   # noqa: BLE001
                err = repr(exc)
            bus = ctypes.create_string_buffer(64)
            if self.lib.ndp_device_pci_bus_id(bus, 64) != 0:
                bus.value = b"?"
            handles = [None] * self.world
            dist.all_gather_object(handles, (self.rank, os.getpid(), handle.raw, err, bus.value.decode()), group=group)
            # ranks that report the same PCI bus id share a GPU (tests, mis-set LOCAL_RANK): the hand-shake then
            # depends on the scheduler keeping every rank's kernels resident at once, see shared_device
            ids = [h[4] for h in handles]
            self.shared_device = len(set(ids)) < len(ids)
            self.bus_ids = ids
            self.struct = _capi.P2P()
            self.struct.world, self.struct.rank, self.struct.timeout_ms = self.world, self.rank, int(timeout_ms)
            if err is None and all(h[3] is None for h in handles):
                try:
                    for r, (src, pid, raw, _e, _bus) in enumerate(handles):
                        if src != r:
                            raise RuntimeError("peer-to-peer exchange: handle list out of rank order")
                        if r == self.rank:
                            self.struct.region[r] = self._region.value
                            continue
                        if pid == os.getpid():
                            raise RuntimeError("peer-to-peer exchange needs one PROCESS per rank (hipIpc)")
                        mapped = ctypes.c_void_p()
                        _capi.check(self.lib.ndp_p2p_open(ctypes.create_string_buffer(raw, len(raw)), ctypes.byref(mapped)),
                                    "ndp_p2p_open(rank %d)" % r)
                        self._mapped.append(mapped)
                        self.struct.region[r] = mapped.value
                except Exception as exc:                               # noqa: BLE001
                    err = repr(exc)
            else:
                err = err or "a peer failed: %s" % next(h[3] for h in handles if h[3] is not None)
            errs = [None] * self.world
            dist.all_gather_object(errs, err, group=group)
        if any(e is not None for e in errs):
            self._release_local()
            raise RuntimeError("peer-to-peer exchange could not be set up: %s" % next(e for e in errs if e is not None))
        self._barrier()

    # ---- helpers
    def _release_local(self):
        with torch.cuda.device(self.device):
            for m in self._mapped:
                self.lib.ndp_p2p_close(m)
            self._mapped = []
            if self._region is not None and self._region.value:
                self.lib.ndp_p2p_region_free(self._region)
        self._region = None

    def _barrier(self):
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)

    def pointer(self):
        import ctypes
        return ctypes.pointer(self.struct)

    def status(self):
        """0 = ok; 1 + r = a wait for rank r's push timed out (results since then are garbage)."""
        import ctypes
        out = ctypes.c_int32(-1)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            self._capi.check(self.lib.ndp_p2p_status(self.pointer(), ctypes.byref(out)), "ndp_p2p_status")
        return int(out.value)

    def diagnostics(self):
        """The record the first waiter that gave up left behind (include/ndp.h, ndp_p2p_diagnostics) or None."""
        import ctypes
        words = (ctypes.c_int32 * self._capi.P2P_DIAG_WORDS)()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            self._capi.check(self.lib.ndp_p2p_diagnostics(self.pointer(), words), "ndp_p2p_diagnostics")
        if words[0] == 0:
            return None
        return {"peer": words[0] - 1, "workgroup": words[1], "net": "G" if words[2] else "D", "expected_step": words[3],
                "flag_seen": words[4], "waited_ms": (words[6] & 0xffffffff) / 1e5}

    def _raise_timeout(self):
        d = self.diagnostics() or {}
        raise RuntimeError(
            "peer-to-peer gradient exchange: rank %d timed out waiting for rank %s (net %s, workgroup %s: expected "
            "step %s, saw flag %s after %.0f ms)%s"
            % (self.rank, d.get("peer", "?"), d.get("net", "?"), d.get("workgroup", "?"), d.get("expected_step", "?"),
               d.get("flag_seen", "?"), d.get("waited_ms", float("nan")),
               "; ranks share a GPU (%s): the hand-shake needs every rank's kernels resident at the same time"
               % ", ".join(self.bus_ids) if self.shared_device else ""))

    def check(self):
        if self.status() != 0:
            self._raise_timeout()

    def poll(self):
        """Free early-abort check for a training loop, called once per (graph) launch: enqueue an asynchronous copy
        of the status word behind the work launched so far, and raise if the copy enqueued by an EARLIER call --
        long complete -- already shows a timed-out wait.  Never synchronises."""
        import ctypes
        if self._status_host is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        if int(self._status_host[0]) != 0:
            self._raise_timeout()
        with torch.cuda.device(self.device):
            self._capi.check(self.lib.ndp_p2p_status_async(self.pointer(), ctypes.c_void_p(self._status_host.data_ptr()),
                                                           self._capi.stream_ptr()), "ndp_p2p_status_async")

    def reset(self):
        """Zero flags and status (collective: no exchange may be in flight on any rank)."""
        self._barrier()
        with torch.cuda.device(self.device):
            self._capi.check(self.lib.ndp_p2p_region_reset(self._region), "ndp_p2p_region_reset")
        if self._status_host is not None:
            self._status_host.zero_()
        self._barrier()

    def all_reduce(self, x, step_word, net=0, out=None):
        """out = sum over ranks of x (flat fp32 CUDA tensor); step_word: int32 CUDA tensor whose
        [0] is this exchange's number (same on all ranks, increasing per net)."""
        c = self._capi
        c.require_gpu_f32(x, "x")
        out = torch.empty_like(x) if out is None else out
        with torch.cuda.device(self.device):
            c.check(self.lib.ndp_p2p_all_reduce(self.pointer(), int(net), c.ptr(x), c.ptr(out), x.numel(),
                                                c.ptr(step_word), c.stream_ptr()), "ndp_p2p_all_reduce")
        return out

    def self_check(self, trials=6, check_timeout_ms=200):
        """Exchange integer-valued vectors (their sums are exact in any order) on both nets and
        compare with torch.distributed's all-reduce.  Collective; True on EVERY rank only if all
        ranks saw exact results and no timeout.  Leaves the region reset."""
        ok = True
        # a node on which the exchange does not work must cost milliseconds, not timeout_ms x waits: the check runs
        # with its own short bound (the ranks enter it together, behind a barrier)
        saved_timeout, self.struct.timeout_ms = self.struct.timeout_ms, int(check_timeout_ms)
        try:
            with torch.cuda.device(self.device):
                # load the library's code object on this device first (the first launch of a process pays for it)
                warm = torch.empty(4, dtype=torch.float32, device=self.device)
                self._capi.check(self.lib.ndp_uniform_noise(self._capi.ptr(warm), 4, 0, None, self._capi.stream_ptr()),
                                 "warm-up")
                self._barrier()
                word = torch.zeros(4, dtype=torch.int32, device=self.device)
                for net, n in ((0, 58305), (1, 83780)):
                    for trial in range(1, trials + 1):
                        gen = torch.Generator().manual_seed(1000 * trial + 10 * self.rank + net)
                        x = torch.randint(-4096, 4096, (n,), generator=gen).float().to(self.device)
                        word.fill_(trial)
                        got = self.all_reduce(x, word, net=net)
                        ref = x.clone()
                        if dist.get_backend(self.group) == "nccl":
                            dist.all_reduce(ref, group=self.group)
                        else:
                            ref_cpu = ref.cpu()
                            dist.all_reduce(ref_cpu, group=self.group)
                            ref = ref_cpu.to(self.device)
                        ok = ok and bool(torch.equal(got, ref))
                ok = ok and self.status() == 0
        except Exception as exc:                             # noqa: BLE001 - any failure disables the path
            import sys
            print("ndivplanning_amd: peer-to-peer self-check raised %r" % (exc,), file=sys.stderr)
            ok = False
        self.struct.timeout_ms = saved_timeout
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend(self.group) == "nccl":
            flag = flag.to(self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        self.reset()
        return bool(flag.item() == 1)

    def measure(self, n=83780, iters=40):
        """(p2p_us, collective_us): average time of one all-reduce of n floats through this exchange and through
        torch.distributed (RCCL), GPU-timed on this rank, maximum over ranks.  Collective; resets the region."""
        with torch.cuda.device(self.device):
            x = torch.ones(n, device=self.device)
            out = torch.empty_like(x)
            words = torch.arange(1, iters + 9, dtype=torch.int32, device=self.device).repeat_interleave(4).reshape(-1, 4)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for i in range(8):                                   # warm-up
                self.all_reduce(x, words[i], net=1, out=out)
            self._barrier()
            ev[0].record()
            for i in range(8, 8 + iters):
                self.all_reduce(x, words[i], net=1, out=out)
            ev[1].record()
            nccl = dist.get_backend(self.group) == "nccl"
            y = x.clone() if nccl else x.cpu()
            for _ in range(4):
                dist.all_reduce(y, group=self.group)
            self._barrier()
            ev[2].record()
            for _ in range(iters):
                dist.all_reduce(y, group=self.group)
            ev[3].record()
            torch.cuda.synchronize(self.device)
            t = torch.tensor([ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])], dtype=torch.float64)
            t = t * 1e3 / iters
            if nccl:
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.reset()
        return float(t[0]), float(t[1])

    def close(self):
        if self._region is None:
            return
        self._barrier()
        with torch.cuda.device(self.device):
            for m in self._mapped:
                self.lib.ndp_p2p_close(m)
            self._mapped = []
            self._barrier()
            self.lib.ndp_p2p_region_free(self._region)
        self._region = None


last_exchange_report = {}       # filled by make_exchange: measured all-reduce times on this node


def measure_collective(device, n=83780, iters=40, group=None):
    """Average GPU time (us, maximum over ranks) of one torch.distributed SUM all-reduce of n floats."""
    device = torch.device(device)
    nccl = dist.get_backend(group) == "nccl"
    with torch.cuda.device(device):
        y = torch.ones(n, device=device) if nccl else torch.ones(n)
        for _ in range(4):
            dist.all_reduce(y, group=group)
        torch.cuda.synchronize(device)
        dist.barrier(group=group)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = __import__("time").perf_counter()
        ev[0].record()
        for _ in range(iters):
            dist.all_reduce(y, group=group)
        ev[1].record()
        torch.cuda.synchronize(device)
        wall = (__import__("time").perf_counter() - t0) * 1e6 / iters
        us = ev[0].elapsed_time(ev[1]) * 1e3 / iters if nccl else wall
        t = torch.tensor([us], dtype=torch.float64, device=device if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0])


def make_exchange(device, world, log=None):
    """Pick the gradient exchange of a data-parallel run: the in-kernel peer-to-peer exchange when
    it passes its self-check on this node, torch.distributed's all-reduce (RCCL) otherwise.
    NDP_DP_EXCHANGE=rccl|p2p forces one.  Returns (p2p_or_None, reduce_fn_or_None, name).  In "auto" the
    exchange is also timed against the collective (the figures go to `log`): the in-kernel exchange additionally
    saves two kernels and the eager launches of the step, so it is kept unless it is slower by more than
    NDP_P2P_MARGIN_US (default 15) per all-reduce."""
    if world <= 1:
        return None, None, "none"
    want = os.environ.get("NDP_DP_EXCHANGE", "auto")
    if want != "rccl" and world <= 8 and torch.device(device).type == "cuda":
        p2p = None
        try:
            p2p = P2PExchange(device)
            good = True
            if p2p.shared_device and want != "p2p":
                # Precondition of the in-kernel hand-shake: every rank's reduce kernel is resident while the others
                # wait for its flags.  One GPU per rank guarantees that; ranks SHARING a GPU depend on how the
                # hardware scheduler interleaves the processes' queues (round 1: 4 ranks on one GPU timed out in
                # 1 run of 3) -- so the collective is used there unless NDP_DP_EXCHANGE=p2p asks for it (tests).
                good = False
                if log:
                    log("ranks share a GPU (%s): using the collective exchange" % ", ".join(p2p.bus_ids))
            # (forced onto a shared GPU -- the tests -- the ranks' kernels time-slice: keep the long bound there)
            good = good and p2p.self_check(check_timeout_ms=p2p.struct.timeout_ms if p2p.shared_device else 200)
            if good:
                t_p2p, t_coll = p2p.measure()
                last_exchange_report.update(p2p_us=round(t_p2p, 2), collective_us=round(t_coll, 2), floats=83780)
                if log:
                    log("all-reduce of 83,780 floats over %d ranks: peer-to-peer %.1f us, torch.distributed %.1f us"
                        % (world, t_p2p, t_coll))
                if want == "auto" and t_p2p > t_coll + float(os.environ.get("NDP_P2P_MARGIN_US", "15")):
                    good = False
        except Exception as exc:                                 # noqa: BLE001
            good = False
            if log:
                log("peer-to-peer exchange unavailable: %r" % (exc,))
        # every rank must take the same branch: self_check already agreed; constructor failures are
        # agreed on here
        flag = torch.tensor([1 if good else 0], dtype=torch.int32,
                            device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 1:
            return p2p, None, "p2p"
        if p2p is not None:
            try:
                p2p.close()                                      # collective: every rank is on this branch
            except Exception:                                    # noqa: BLE001
                pass
        if want == "p2p":
            raise RuntimeError("NDP_DP_EXCHANGE=p2p but the peer-to-peer exchange failed its self-check")
        if log:
            log("peer-to-peer exchange failed its self-check; using the RCCL all-reduce")
    if world > 1 and "collective_us" not in last_exchange_report and torch.device(device).type == "cuda":
        try:
            last_exchange_report.update(collective_us=round(measure_collective(device), 2), floats=83780)
        except Exception as exc:                                 # noqa: BLE001 - a report, not a requirement
            if log:
                log("could not time the collective: %r" % (exc,))
    return None, sum_all_reduce(), "rccl"
