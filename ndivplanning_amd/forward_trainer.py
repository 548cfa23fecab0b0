"""One iteration of the reference's forward-model training loop (train_forward_model.py:98-112) as gfx950 kernels:

    state_fut_resid_hat = forward_autoencoder(state_cur, action)      ->  ndp_fm_train_grads   (forward, training-mode
    loss = mse(state_fut_resid_hat, state_fut - state_cur)                BatchNorm, MSE, backward: every gradient)
    optimizer.zero_grad(); loss.backward()
    optimizer.step()                                                   ->  ndp_fm_apply_adam

The trainer owns the flat parameter vector the kernels read (include/ndp.h: forward model), its gradient, Adam moments
and the running BatchNorm statistics; `sync_to_module()` writes them back into the `ForwardAutoencoder` (the reference
saves the whole module: train_forward_model.py:157-163).  `reduce_fn(grad)` -- when given, called between the two
library calls -- is where a data-parallel driver all-reduces the gradient ("same DP recipe";
the loss is a mean over the local batch, so the driver averages).  `bucket_reduce` (dp.BucketedMeanAllReduce) does the
same per gradient bucket on a communication stream, each bucket as soon as the backward pass has completed it
(ndp_fm_grad_buckets / ndp_fm_bucket_wait): the all-reduce runs beside the rest of the backward pass."""
import torch

from . import _capi
from .models import forward_encoder as FE


class ForwardModelTrainer:
    def __init__(self, model: FE.ForwardAutoencoder, batch: int, lr: float = 2e-4, betas=(0.5, 0.999), eps: float = 1e-8,
                 reduce_fn=None, keep_residual: bool = False, bucket_reduce=None, sync_batchnorm_world: int = 1):
        self.lib = _capi.load()
        self.model = model
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _capi.NdpError("ForwardModelTrainer needs the model on a ROCm GPU (got %s); there is no CPU path" % dev)
        self.device, self.batch = dev, int(batch)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        if reduce_fn is not None and bucket_reduce is not None:
            raise ValueError("give either reduce_fn (one collective) or bucket_reduce (per-bucket, overlapped)")
        self.reduce_fn, self.bucket_reduce = reduce_fn, bucket_reduce
        f32 = dict(dtype=torch.float32, device=dev)
        self.params, self.stats = FE.pack_module(model, dev)
        self.grad = torch.zeros_like(self.params)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.step_word = torch.zeros(4, dtype=torch.int32, device=dev)      # Adam state word (include/ndp.h)
        self.loss = torch.zeros(1, **f32)
        self.loss_sum = torch.zeros(1, **f32)
        self.resid = torch.zeros(self.batch, 3, 128, 128, **f32) if keep_residual else None
        self.workspace = torch.empty(self.lib.ndp_fm_workspace_floats(self.batch), **f32)
        self.steps = 0
        # BatchNorm statistics over all ranks' images (dp.CrossRankBatchNorm): W ranks x B / W images = one process x B
        self.stat_sync = None
        if int(sync_batchnorm_world) > 1:
            from . import dp
            self.stat_sync = dp.CrossRankBatchNorm(self.workspace, int(sync_batchnorm_world))
        with torch.cuda.device(dev):
            _capi.check(self.lib.ndp_fm_pack_params(_capi.ptr(self.params), _capi.ptr(self.workspace),
                                                    _capi.stream_ptr(dev)), "ndp_fm_pack_params")

    def _check(self, t, shape, what, dtype=torch.float32):
        if t.device != self.device or t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous():
            raise _capi.NdpError("%s: expected a contiguous %s %s tensor on %s, got %s %s on %s"
                                 % (what, dtype, list(shape), self.device, t.dtype, list(t.shape), t.device))

    def grads(self, state_cur, state_fut, actions):
        """forward + loss + backward: fills .grad, .loss (device scalar, the reference's `loss`), adds it to .loss_sum.
        Frames: the reference's float tensors [n,3,128,128] in [-1,1], or decoded camera frames as bytes [n,128,128,3]
        (normalised by the kernels as they read them: utils/hdf5_load.py:9-11's formula, a quarter of the upload)."""
        n = int(state_cur.shape[0])
        if not 1 <= n <= self.batch:
            raise _capi.NdpError("batch of %d images with a trainer built for at most %d" % (n, self.batch))
        u8 = state_cur.dtype == torch.uint8
        shape = (n, 128, 128, 3) if u8 else (n, 3, 128, 128)
        self._check(state_cur, shape, "state_cur", state_cur.dtype if u8 else torch.float32)
        self._check(state_fut, shape, "state_fut", torch.uint8 if u8 else torch.float32)
        self._check(actions, (n, 4), "actions")
        p = _capi.ptr
        fn, name = ((self.lib.ndp_fm_train_grads_u8, "ndp_fm_train_grads_u8") if u8
                    else (self.lib.ndp_fm_train_grads, "ndp_fm_train_grads"))
        with torch.cuda.device(self.device):
            _capi.check(fn(p(self.params), p(self.stats), p(state_cur), p(state_fut), p(actions), n,
                           p(self.grad), p(self.loss), p(self.loss_sum), p(self.resid) if self.resid is not None else None,
                           p(self.workspace), _capi.stream_ptr(self.device)), name)
        if self.stat_sync is not None:
            self.stat_sync.check()
        return self.loss

    def apply(self):
        """optimizer.step()"""
        p = _capi.ptr
        with torch.cuda.device(self.device):
            _capi.check(self.lib.ndp_fm_apply_adam(p(self.params), p(self.grad), p(self.exp_avg), p(self.exp_avg_sq),
                                                   p(self.step_word), self.lr, self.betas[0], self.betas[1], self.eps,
                                                   p(self.workspace), _capi.stream_ptr(self.device)), "ndp_fm_apply_adam")
        self.steps += 1

    def step(self, state_cur, state_fut, actions):
        """The loop body of train_forward_model.py:98-112 for one frame pair; returns the loss (device scalar)."""
        self.grads(state_cur, state_fut, actions)
        if self.bucket_reduce is not None:
            self.bucket_reduce(self.grad, self.device)
        elif self.reduce_fn is not None:
            self.reduce_fn(self.grad)
        self.apply()
        return self.loss

    def close(self):
        """Remove the process-wide cross-rank statistics hook this trainer installed (if any)."""
        if self.stat_sync is not None:
            self.stat_sync.close()
            self.stat_sync = None

    def gradient_buckets(self):
        """[(offset, count)] of the flat gradient, in the order the backward pass completes them."""
        return _capi.fm_grad_buckets()

    # intermediate maps of the last grads() call (tests, inspection): name -> (workspace tensor index, side, channels kept)
    _MAPS = {"feat1": (1, 64, 128, 64, 128), "up5": (1, 64, 128, 0, 64), "feat2": (2, 32, 256, 128, 256), "up4": (2, 32, 256, 0, 128),
             "feat3": (3, 16, 512, 256, 512), "up3": (3, 16, 512, 0, 256), "feat4": (4, 8, 1024, 512, 1024),
             "up2": (4, 8, 1024, 0, 512), "feat5": (5, 4, 2048, 1024, 2048), "up1": (5, 4, 2048, 0, 1024),
             "up6": (16, 128, 32, 0, 32), "r1": (18, 128, 16, 0, 16)}

    def activation(self, name, n):
        """Post-ReLU map `name` (feat1..5, up1..6, r1) of the last call on n images, NCHW."""
        idx, side, ld, c0, c1 = self._MAPS[name]
        off = self.lib.ndp_fm_workspace_offset(n, idx)
        view = self.workspace[off:off + n * side * side * ld].view(n, side, side, ld)
        return view[..., c0:c1].permute(0, 3, 1, 2).contiguous()

    def load_from_module(self):
        """Take parameters and running statistics from the module again (after they were changed from outside)."""
        self.params, self.stats = FE.pack_module(self.model, self.device)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.ndp_fm_pack_params(_capi.ptr(self.params), _capi.ptr(self.workspace),
                                                    _capi.stream_ptr(self.device)), "ndp_fm_pack_params")

    def sync_to_module(self):
        FE.unpack_into_module(self.model, self.params, self.stats, batches_tracked=self.steps)
        return self.model

    def named_gradients(self):
        """name -> gradient in the module's own tensor shapes (tests, inspection)."""
        return FE.unpack_vector(self.grad, self.model)

    def named_parameters(self):
        return FE.unpack_vector(self.params, self.model)
