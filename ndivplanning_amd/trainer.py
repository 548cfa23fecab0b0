"""Fused GAN train step -- the hot loop of the reference's `train_gan.train`
(train_gan.py:159-203) as a handful of gfx950 kernels per step, optionally
captured in a HIP graph, optionally data-parallel.

One step = `discrim_steps` x phase A (`ndp_step_d_grads`: [G forward,] D(real),
D(fake), BCE, D backward, D Adam) + phase B (`ndp_step_g_grads`: D(fake) with the
updated D, G loss, NDiv, backward through D and G, G Adam).  See include/ndp.h.

Data parallelism: each rank holds `flat` of the
`flat_global` rows.  BCE is a mean over the global row count (inv_m_global),
NDiv is a sum, so per-rank gradients are SUMMED by `reduce_fn` (an all-reduce over
RCCL), after which every rank applies the same Adam update to its replica -- or, with
`p2p` (dp.P2PExchange), summed inside the slab-reduce kernels over hipIpc-mapped peer memory, so
the data-parallel step is the single-GPU step (fused Adam, one HIP graph).
"""
import ctypes

import torch

from . import _capi, dp
from .models.gan import ACTION_DIM, CODE_DIM, Decoder, Discriminator


def _on_own_device(fn):
    """Run a GanTrainer method with the trainer's device current: launches, kernel attributes and graph capture
    all act on the current device, which need not be the one the networks live on (reference default
    `gpu_id: 1`, no set_device)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        with torch.cuda.device(self.device):
            return fn(self, *args, **kw)
    return wrapper


class GanTrainer:
    def __init__(self, decoder: Decoder, discriminator: Discriminator, *, flat: int, num_sample: int,
                 lr: float = 2e-4, betas=(0.5, 0.999), eps: float = 1e-8, pairwise_div_factor: float = 0.1,
                 discrim_steps: int = 1, flat_global: int = None, reduce_fn=None, use_graph: bool = True,
                 noise_seed: int = 0, steps_per_launch: int = 1, p2p=None, copy_stream=None):
        self.lib = _capi.load()
        self.decoder, self.discriminator = decoder, discriminator
        self.noise_dim = decoder.noise_dim
        self.flat, self.k = int(flat), int(num_sample)
        self.flat_global = int(flat_global) if flat_global is not None else self.flat
        self.m = self.flat * self.k
        self.discrim_steps = int(discrim_steps)
        if p2p is not None and reduce_fn is not None:
            raise ValueError("give either p2p (in-kernel exchange) or reduce_fn (collective between the phases)")
        self.reduce_fn = reduce_fn
        # dp.P2PExchange: the slab-reduce kernels sum the gradients over ranks themselves, the step
        # keeps the single-GPU launch sequence (fused Adam, one graph)
        self.p2p = p2p
        # With a collective between the phases the step cannot be one graph, and replaying
        # three graph segments costs more (~8 us of launch gap each) than launching the 8 kernels
        # eagerly, whose host cost hides behind the GPU work (measured: 136 vs 156 us/step).
        self.use_graph = bool(use_graph) and reduce_fn is None
        self.noise_seed = int(noise_seed)
        # A graph replay costs ~8 us of GPU idle time between replays (measured: the gap between
        # the last kernel of one replay and the first of the next); `steps_per_launch` > 1
        # captures that many consecutive iterations, each with its own input slot, in one graph.
        self.nslots = max(1, int(steps_per_launch))
        self.g_flat = decoder.flat_parameters()
        self.d_flat = discriminator.flat_parameters()
        dev = self.g_flat.device
        if dev.type != "cuda":
            raise _capi.NdpError("GanTrainer needs the networks on a ROCm GPU (got %s); there is no CPU path" % dev)
        self.device = dev
        f32 = dict(dtype=torch.float32, device=dev)
        self.g_grad, self.d_grad = torch.zeros_like(self.g_flat), torch.zeros_like(self.d_flat)
        self.g_m, self.g_v = torch.zeros_like(self.g_flat), torch.zeros_like(self.g_flat)
        self.d_m, self.d_v = torch.zeros_like(self.d_flat), torch.zeros_like(self.d_flat)
        # Adam state words (include/ndp.h): [0] = updates applied so far, [1..3] library scratch
        self.g_step = torch.zeros(4, dtype=torch.int32, device=dev)
        self.d_step = torch.zeros(4, dtype=torch.int32, device=dev)
        self.losses_dev = torch.zeros(4, **f32)
        self.loss_sums = torch.zeros(4, **f32)
        mpad = _capi.pad_rows(self.m)
        self.action_hat = torch.zeros(mpad, ACTION_DIM, **f32)
        # static input buffers (graph replay needs fixed addresses), one slot per captured step;
        # .codes / .actions / .noise are slot 0
        self.codes_slots = torch.zeros(self.nslots, self.flat, CODE_DIM, **f32)
        self.actions_slots = torch.zeros(self.nslots, self.flat, ACTION_DIM, **f32)
        self.noise_slots = torch.zeros(self.nslots, self.flat, self.k, self.noise_dim, **f32)
        self.codes, self.actions, self.noise = self.codes_slots[0], self.actions_slots[0], self.noise_slots[0]
        self.cfg = _capi.StepConfig(
            noise_dim=self.noise_dim, num_sample=self.k, flat=self.flat,
            inv_m_global=1.0 / float(self.flat_global * self.k), pairwise_div_factor=float(pairwise_div_factor),
            lr=float(lr), beta1=float(betas[0]), beta2=float(betas[1]), eps=float(eps),
            fuse_adam=0 if reduce_fn is not None else 1, device_noise=0, noise_seed=self.noise_seed,
            p2p=p2p.pointer() if p2p is not None else None)
        nws = self.lib.ndp_step_workspace_floats(ctypes.byref(self.cfg))
        if nws <= 0:
            raise _capi.NdpError("bad step configuration (flat=%d, num_sample=%d)" % (self.flat, self.k))
        self.workspace = torch.empty(nws, **f32)
        self._graphs = None
        # step_many_from_host uploads on this stream (default: one of its own).  A process should keep the number of
        # streams it really uses small: beyond ~4 hardware queues the GPU time-slices them instead of running them side
        # by side (measured: every kernel of a two-stream workload 2-4x slower once two more streams had been used)
        self._copy_stream = copy_stream
        self._stage = None
        self._alt_slots = None
        self._device_noise_now = False
        self._d_calls = 0
        with torch.cuda.device(self.device):
            self._bind()

    # -- plumbing ---------------------------------------------------------------
    def _versions(self):
        # in-place writes through torch bump the version counter of the tensor they go through:
        # the flat buffers and every parameter view (which keeps a counter of its own)
        return (self.g_flat._version, self.d_flat._version,
                tuple(p._version for p in self.decoder._param_list()),
                tuple(p._version for p in self.discriminator._param_list()))

    def _repack(self):
        """Rebuild the lane-ordered weight copies the step kernels read (include/ndp.h)."""
        _capi.check(self.lib.ndp_step_pack_params(ctypes.byref(self.cfg), ctypes.byref(self.buf), _capi.stream_ptr(self.device)),
                    "ndp_step_pack_params")
        self._seen_versions = self._versions()

    def _bind(self):
        self.g_flat = self.decoder.flat_parameters()
        self.d_flat = self.discriminator.flat_parameters()
        p = _capi.ptr
        self.buf = _capi.StepBuffers(
            g_params=p(self.g_flat), g_grad=p(self.g_grad), g_exp_avg=p(self.g_m), g_exp_avg_sq=p(self.g_v),
            d_params=p(self.d_flat), d_grad=p(self.d_grad), d_exp_avg=p(self.d_m), d_exp_avg_sq=p(self.d_v),
            g_step=p(self.g_step), d_step=p(self.d_step), losses=p(self.losses_dev), loss_sums=p(self.loss_sums),
            action_hat=p(self.action_hat), workspace=p(self.workspace))
        # the reference adds the LAST D loss of an iteration to its epoch sum (train_gan.py:205):
        # earlier D steps of a discrim_steps_per_gen > 1 iteration run without the running sums
        self.buf_nosum = _capi.StepBuffers()
        ctypes.memmove(ctypes.byref(self.buf_nosum), ctypes.byref(self.buf), ctypes.sizeof(_capi.StepBuffers))
        self.buf_nosum.loss_sums = None
        self._repack()

    def _inputs(self, slot, sset=0):
        """(codes, actions, noise) buffers of input slot `slot`; sset 1 = the second slot set that
        step_many_from_host alternates with (allocated on first use)."""
        if sset == 0:
            return self.codes_slots[slot], self.actions_slots[slot], self.noise_slots[slot]
        if self._alt_slots is None:
            self._alt_slots = (torch.zeros_like(self.codes_slots), torch.zeros_like(self.actions_slots),
                               torch.zeros_like(self.noise_slots))
        return tuple(t[slot] for t in self._alt_slots)

    def _phase_a(self, first, device_noise=False, last=True, slot=0, sset=0):
        # device noise: the G forward kernel draws U[0,1) itself and fills the slot's noise buffer
        self.cfg.device_noise = 1 if device_noise else 0
        buf = self.buf if last else self.buf_nosum
        codes, actions, noise = self._inputs(slot, sset)
        _capi.check(self.lib.ndp_step_d_grads(ctypes.byref(self.cfg), ctypes.byref(buf), _capi.ptr(codes),
                                              _capi.ptr(actions), _capi.ptr(noise), 1 if first else 0,
                                              _capi.stream_ptr(self.device)), "ndp_step_d_grads")

    def _phase_b(self, slot=0, sset=0):
        codes, actions, noise = self._inputs(slot, sset)
        _capi.check(self.lib.ndp_step_g_grads(ctypes.byref(self.cfg), ctypes.byref(self.buf), _capi.ptr(codes),
                                              _capi.ptr(actions), _capi.ptr(noise), _capi.stream_ptr(self.device)),
                    "ndp_step_g_grads")

    # the step as a list of segments; between segments the data-parallel driver
    # all-reduces the gradient the previous segment produced
    def _segments(self, device_noise, slot=0, sset=0):
        segs = []

        def seg_d(first, last):
            def run():
                self._phase_a(first, device_noise, last, slot, sset)
            return run

        def seg_g():
            self._phase_b(slot, sset)

        def d_update():
            _capi.check(self.lib.ndp_step_apply_adam(ctypes.byref(self.cfg), ctypes.byref(self.buf), 0,
                                                     _capi.stream_ptr(self.device)), "ndp_step_apply_adam")

        def g_update():
            _capi.check(self.lib.ndp_step_apply_adam(ctypes.byref(self.cfg), ctypes.byref(self.buf), 1,
                                                     _capi.stream_ptr(self.device)), "ndp_step_apply_adam")

        fused = self.reduce_fn is None
        for it in range(self.discrim_steps):
            segs.append((seg_d(it == 0, it == self.discrim_steps - 1), None if fused else self.d_grad))
            if not fused:
                segs.append((d_update, None))
        segs.append((seg_g, None if fused else self.g_grad))
        if not fused:
            segs.append((g_update, None))
        return segs

    # dp.run_step backend protocol (non-fused mode)
    @_on_own_device
    def d_grads(self, first):
        self._d_calls = 0 if first else self._d_calls + 1
        self._phase_a(first, self._device_noise_now, self._d_calls == self.discrim_steps - 1)
        return self.d_grad

    @_on_own_device
    def apply_d(self, grad):
        _capi.check(self.lib.ndp_step_apply_adam(ctypes.byref(self.cfg), ctypes.byref(self.buf), 0,
                                                 _capi.stream_ptr(self.device)), "ndp_step_apply_adam")

    @_on_own_device
    def g_grads(self):
        self._phase_b()
        return self.g_grad

    @_on_own_device
    def apply_g(self, grad):
        _capi.check(self.lib.ndp_step_apply_adam(ctypes.byref(self.cfg), ctypes.byref(self.buf), 1,
                                                 _capi.stream_ptr(self.device)), "ndp_step_apply_adam")

    def _run_eager(self, device_noise):
        if self.reduce_fn is not None:
            self._device_noise_now = device_noise
            dp.run_step(self, self.reduce_fn, self.discrim_steps)
            return
        for fn, _ in self._segments(device_noise):
            fn()
        if self.p2p is not None:
            self.p2p.poll()

    def _build_graphs(self, device_noise, nsteps=1, sset=0):
        """Capture maximal runs of segments that need no collective in between; `nsteps`
        consecutive iterations (input slots 0..nsteps-1) when there is no collective at all."""
        # load the code object / set kernel attributes outside of capture
        tmp = torch.empty(4, dtype=torch.float32, device=self.device)
        _capi.check(self.lib.ndp_uniform_noise(_capi.ptr(tmp), 4, 0, None, _capi.stream_ptr(self.device)), "warm-up")
        torch.cuda.synchronize(self.device)
        plan, run = [], []
        all_segments = [sg for slot in range(nsteps) for sg in self._segments(device_noise, slot, sset)]
        for fn, grad in all_segments:
            run.append(fn)
            if grad is not None:
                plan.append((run, grad))
                run = []
        if run:
            plan.append((run, None))
        graphs = []
        for fns, grad in plan:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for fn in fns:
                    fn()
            graphs.append((g, grad))
        return graphs

    # -- public -----------------------------------------------------------------
    @_on_own_device
    def step(self, codes=None, actions=None, noise=None):
        """One training iteration.  `codes` [flat,256], `actions` [flat,4]: this rank's
        shard (None = keep the buffers' current contents); `noise` [flat,K,nz] or None to
        draw U[0,1) on the device."""
        self._check_bindings()
        if codes is not None:
            self.codes.copy_(codes.reshape(self.flat, CODE_DIM), non_blocking=True)
        if actions is not None:
            self.actions.copy_(actions.reshape(self.flat, ACTION_DIM), non_blocking=True)
        device_noise = noise is None
        if not device_noise:
            self.noise.copy_(noise.reshape(self.flat, self.k, self.noise_dim), non_blocking=True)
        if not self.use_graph:
            self._run_eager(device_noise)
            return
        self._replay(device_noise, 1)

    def _replay(self, device_noise, nsteps, sset=0):
        key = (bool(device_noise), nsteps, sset)
        if self._graphs is None:
            self._graphs = {}
        if key not in self._graphs:
            # capture performs no work; the first replay below is the step itself
            self._graphs[key] = self._build_graphs(device_noise, nsteps, sset)
        for g, grad in self._graphs[key]:
            g.replay()
            if grad is not None:
                self.reduce_fn(grad)
        if self.p2p is not None:
            self.p2p.poll()          # a timed-out wait of an EARLIER launch aborts here (no synchronisation)

    @_on_own_device
    def step_many(self, codes=None, actions=None, noise=None, count=None):
        """`steps_per_launch` consecutive iterations in ONE graph replay.  codes [n,flat,256],
        actions [n,flat,4], noise [n,flat,K,nz] (n = steps_per_launch); None keeps what the slots
        hold (codes/actions) or draws device noise.  `count` < steps_per_launch: only the first
        `count` slots (a graph of its own per count).  With a collective between the phases
        (data parallel) or without graphs this is a plain loop over the slots."""
        self._check_bindings()
        n = self.nslots if count is None else int(count)
        if not 1 <= n <= self.nslots:
            raise ValueError("count=%r outside 1..%d" % (count, self.nslots))
        if codes is not None:
            self.codes_slots[:n].copy_(codes.reshape(n, self.flat, CODE_DIM), non_blocking=True)
        if actions is not None:
            self.actions_slots[:n].copy_(actions.reshape(n, self.flat, ACTION_DIM), non_blocking=True)
        device_noise = noise is None
        if not device_noise:
            self.noise_slots[:n].copy_(noise.reshape(n, self.flat, self.k, self.noise_dim), non_blocking=True)
        if self.use_graph and self.reduce_fn is None:
            self._replay(device_noise, n)
            return
        for slot in range(n):
            for fn, grad in self._segments(device_noise, slot):
                fn()
                if grad is not None:
                    self.reduce_fn(grad)

    @_on_own_device
    def step_many_from_host(self, codes_host, actions_host):
        """`steps_per_launch` iterations on a FRESH batch per slot that still sits in pinned host memory
        (codes [n,flat,256], actions [n,flat,4]): the reference uploads every batch (train_gan.py:119-124).
        Two sets of input slots alternate, each with its own captured graph: while the graph of one set computes,
        the copy stream uploads the next launch's batches straight into the other set, and the launch stream only
        waits for that upload's event before the replay -- nothing but the wait sits between two replays.
        Device noise."""
        self._check_bindings()
        n = self.nslots
        if not (codes_host.is_pinned() and actions_host.is_pinned()):
            raise ValueError("step_many_from_host needs pinned host tensors (torch.Tensor.pin_memory())")
        if self._stage is None:
            self._stage = {"stream": self._copy_stream or torch.cuda.Stream(self.device), "next": 0,
                           "uploaded": [torch.cuda.Event(), torch.cuda.Event()],
                           "read": [torch.cuda.Event(), torch.cuda.Event()]}
            # the second slot set is allocated (and zero-filled) on the launch stream: the copy stream must not
            # upload into it before that fill has run (it would be overwritten with zeros when the fill ran later)
            self._inputs(0, 1)
            self._stage["stream"].wait_stream(torch.cuda.current_stream(self.device))
        st = self._stage
        sset = st["next"]
        st["next"] = 1 - sset
        codes = self.codes_slots if sset == 0 else self._alt_slots[0]
        actions = self.actions_slots if sset == 0 else self._alt_slots[1]
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(st["stream"]):
            st["stream"].wait_event(st["read"][sset])    # the last launch that read this set has finished
            codes.copy_(codes_host.reshape(n, self.flat, CODE_DIM), non_blocking=True)
            actions.copy_(actions_host.reshape(n, self.flat, ACTION_DIM), non_blocking=True)
            st["uploaded"][sset].record(st["stream"])
        main.wait_event(st["uploaded"][sset])
        if self.use_graph and self.reduce_fn is None:
            self._replay(True, n, sset)
        else:
            for slot in range(n):
                for fn, grad in self._segments(True, slot, sset):
                    fn()
                    if grad is not None:
                        self.reduce_fn(grad)
        st["read"][sset].record(main)

    def _check_bindings(self):
        if self.g_flat.data_ptr() != self.decoder.flat_parameters().data_ptr() or \
                self.d_flat.data_ptr() != self.discriminator.flat_parameters().data_ptr():
            self._bind()
            self._graphs = None
        elif self._versions() != self._seen_versions:
            # somebody wrote the parameters through torch (load_state_dict, copy_, an optimizer):
            # the kernels' packed copies are stale
            self._repack()

    def losses(self):
        """(D_loss, G_loss, pair_div) of the last step as Python floats (synchronises)."""
        v = self.losses_dev.tolist()
        return v[0], v[1], v[2]

    def pop_loss_sums(self):
        """Running sums of the three losses since the last call (one sync per epoch
        instead of the reference's three per step, train_gan.py:205-207)."""
        v = self.loss_sums.tolist()
        self.loss_sums.zero_()
        return v[0], v[1], v[2]

    @_on_own_device
    def load_adam_state(self, g_state, d_state):
        """Teacher-forcing hook for parity tests: {'m': flat, 'v': flat, 't': int} per network."""
        for (m, v, step), st in (((self.g_m, self.g_v, self.g_step), g_state), ((self.d_m, self.d_v, self.d_step), d_state)):
            m.copy_(st["m"])
            v.copy_(st["v"])
            step.zero_()
            step[0] = int(st["t"])
