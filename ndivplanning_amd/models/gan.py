"""Generator / Discriminator of the action GAN -- mirror of the reference's
`models/gan.py` (`Decoder`, `Discriminator`): same constructors, same `fc*`
`nn.Linear` attributes (so `state_dict()` keys and whole-module pickles are
interchangeable with the reference's), same `forward` signatures, same
`weight_init` (a no-op for Linear layers, models/gan.py:15-18).

`forward` runs the fused gfx950 kernels of libndp_hip.so (include/ndp.h) through
a `torch.autograd.Function`; parameters live in one flat fp32 buffer per network
(the `fc*.weight/bias` parameters are views into it), which is what the kernels
and the fused trainer (`ndivplanning_amd.trainer.GanTrainer`) read and update.
There is no CPU path: calling `forward` on CPU tensors raises.
"""
import torch
import torch.nn as nn

from .. import _capi

CODE_DIM = _capi.CODE_DIM
ACTION_DIM = _capi.ACTION_DIM

# (in, out) per layer -- models/gan.py:67-71 and 94-97
_G_HIDDEN = (128, 64, 128, 256)
_D_HIDDEN = (64, 128, 256)


def normal_init(m, mean, std):
    """models/gan.py:15-18: only (transposed) convolutions are re-initialised."""
    if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
        m.weight.data.normal_(mean, std)
        m.bias.data.zero_()


class _FlatParamsMixin:
    """Keeps every parameter of the module as a view into one contiguous buffer, in
    state_dict order, re-establishing the views whenever `.to()` / `load_state_dict`
    / unpickling replaced the storages."""

    def _layers(self):
        return [getattr(self, "fc%d" % i) for i in range(1, self._n_layers + 1)]

    def _param_list(self):
        out = []
        for layer in self._layers():
            out += [layer.weight, layer.bias]
        return out

    def flat_parameters(self):
        """The flat fp32 parameter vector (fc1.weight, fc1.bias, fc2.weight, ...);
        parameters are (re)bound to views of it if needed."""
        params = self._param_list()
        flat = self.__dict__.get("_flat")
        ok = flat is not None and flat.device == params[0].device
        if ok:
            off = 0
            base = flat.data_ptr()
            for p in params:
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            dev = params[0].device
            total = sum(p.numel() for p in params)
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in params:
                    n = p.numel()
                    flat[off:off + n].copy_(p.detach().reshape(-1).to(torch.float32))
                    p.data = flat[off:off + n].view(p.shape)
                    off += n
            self.__dict__["_flat"] = flat
        return flat

    def __getstate__(self):
        # whole-module pickles (train_gan.py:254-266) must not carry the alias buffer
        state = self.__dict__.copy()
        state.pop("_flat", None)
        return state


class _GForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, module, *params):
        lib = _capi.load()
        flat = module.flat_parameters()
        nz = module.noise_dim
        m, width = z.shape
        ld = z.stride(0)
        need_grad = any(ctx.needs_input_grad[2:])
        acts = _capi.empty(lib.ndp_g_acts_floats(m), z) if need_grad else None
        out = torch.empty((m, ACTION_DIM), dtype=torch.float32, device=z.device)
        noise_view = z[:, CODE_DIM:]
        with _capi.on_device(z):
            _capi.check(lib.ndp_g_forward(_capi.ptr(flat), nz, _capi.ptr(z), ld, 1, _capi.ptr(noise_view), ld, m,
                                          _capi.ptr(acts), _capi.ptr(out), _capi.stream_ptr()), "ndp_g_forward")
        ctx.module, ctx.m = module, m
        if need_grad:
            ctx.save_for_backward(z, acts)
        return out

    @staticmethod
    def backward(ctx, d_action):
        lib = _capi.load()
        module, m = ctx.module, ctx.m
        z, acts = ctx.saved_tensors
        flat = module.flat_parameters()
        nz = module.noise_dim
        ld = z.stride(0)
        d_action = d_action.contiguous()
        grad = torch.empty_like(flat)
        ws = _capi.empty(lib.ndp_g_bwd_ws_floats(m, nz), z)
        noise_view = z[:, CODE_DIM:]
        with _capi.on_device(z):
            _capi.check(lib.ndp_g_backward(_capi.ptr(flat), nz, _capi.ptr(z), ld, 1, _capi.ptr(noise_view), ld, m,
                                           _capi.ptr(acts), _capi.ptr(d_action), _capi.ptr(grad), _capi.ptr(ws),
                                           _capi.stream_ptr()), "ndp_g_backward")
        grads, off = [], 0
        for p in module._param_list():
            n = p.numel()
            grads.append(grad[off:off + n].view(p.shape))
            off += n
        return (None, None) + tuple(grads)


class Decoder(_FlatParamsMixin, nn.Module):
    """G: (code 256 || noise) -> 128 -> 64 -> 128 -> 256 -> 4 actions, ReLU
    (reference models/gan.py:61-86)."""

    _n_layers = 5

    def __init__(self, noise_dim):
        super().__init__()
        self.noise_dim = int(noise_dim)
        widths = (CODE_DIM + self.noise_dim,) + _G_HIDDEN + (ACTION_DIM,)
        for i in range(self._n_layers):
            setattr(self, "fc%d" % (i + 1), nn.Linear(widths[i], widths[i + 1]))

    def __setstate__(self, state):
        self.__dict__.update(state)
        if "noise_dim" not in self.__dict__:          # pickles written by the reference class
            self.noise_dim = self.fc1.in_features - CODE_DIM

    def weight_init(self, mean, std):
        for name in self._modules:
            normal_init(self._modules[name], mean, std)

    def forward(self, z):
        _capi.require_gpu_f32(z, "z")
        if z.dim() != 2 or z.size(1) != CODE_DIM + self.noise_dim:
            raise _capi.NdpError("Decoder expects [M, %d], got %s" % (CODE_DIM + self.noise_dim, tuple(z.shape)))
        if not 1 <= self.noise_dim <= _capi.MAX_NOISE_DIM:
            raise _capi.NdpError("noise_dim=%d outside 1..%d" % (self.noise_dim, _capi.MAX_NOISE_DIM))
        if z.requires_grad:
            raise NotImplementedError("gradient w.r.t. the generator input is not part of the training path "
                                      "(the reference detaches the codes, train_gan.py:152-153)")
        if z.stride(1) != 1:
            z = z.contiguous()
        if self.fc1.weight.device != z.device:
            raise _capi.NdpError("Decoder parameters are on %s, input on %s" % (self.fc1.weight.device, z.device))
        return _GForward.apply(z, self, *self._param_list())


class _DForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, action, code, module, *params):
        lib = _capi.load()
        flat = module.flat_parameters()
        m = action.shape[0]
        logits = torch.empty((m, 1), dtype=torch.float32, device=action.device)
        with _capi.on_device(action):
            _capi.check(lib.ndp_d_forward(_capi.ptr(flat), _capi.ptr(action), 1, _capi.ptr(code), code.stride(0), 1, m,
                                          _capi.ptr(logits), _capi.stream_ptr()), "ndp_d_forward")
        ctx.module, ctx.m = module, m
        ctx.save_for_backward(action, code)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        lib = _capi.load()
        module, m = ctx.module, ctx.m
        action, code = ctx.saved_tensors
        flat = module.flat_parameters()
        want_params = any(ctx.needs_input_grad[3:])
        want_action = ctx.needs_input_grad[0]
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. state_code is not part of the training path "
                                      "(codes are detached, train_gan.py:152-153)")
        d_logits = d_logits.contiguous().view(-1)
        grad = torch.empty_like(flat) if want_params else None
        d_action = torch.empty_like(action) if want_action else None
        ws = _capi.empty(lib.ndp_d_bwd_ws_floats(m), action) if want_params else None
        with _capi.on_device(action):
            _capi.check(lib.ndp_d_backward(_capi.ptr(flat), _capi.ptr(action), 1, _capi.ptr(code), code.stride(0), 1, m,
                                           _capi.ptr(d_logits), _capi.ptr(grad), _capi.ptr(d_action), _capi.ptr(ws),
                                           _capi.stream_ptr()), "ndp_d_backward")
        grads, off = [], 0
        for p in module._param_list():
            n = p.numel()
            grads.append(grad[off:off + n].view(p.shape) if want_params else None)
            off += n
        return (d_action, None, None) + tuple(grads)


class Discriminator(_FlatParamsMixin, nn.Module):
    """D: cat[action 4, code 256] -> 64 -> 128 -> 256 -> 1 logit, LeakyReLU(0.01)
    (reference models/gan.py:89-110)."""

    _n_layers = 4

    def __init__(self):
        super().__init__()
        widths = (CODE_DIM + ACTION_DIM,) + _D_HIDDEN + (1,)
        for i in range(self._n_layers):
            setattr(self, "fc%d" % (i + 1), nn.Linear(widths[i], widths[i + 1]))

    def weight_init(self, mean, std):
        for name in self._modules:
            normal_init(self._modules[name], mean, std)

    def forward(self, action, state_code):
        _capi.require_gpu_f32(action, "action")
        _capi.require_gpu_f32(state_code, "state_code")
        if action.dim() != 2 or action.size(1) != ACTION_DIM or state_code.dim() != 2 \
                or state_code.size(1) != CODE_DIM or state_code.size(0) != action.size(0):
            raise _capi.NdpError("Discriminator expects action [M,4] and state_code [M,256], got %s and %s"
                                 % (tuple(action.shape), tuple(state_code.shape)))
        action = action.contiguous()
        if state_code.stride(1) != 1:
            state_code = state_code.contiguous()
        if not (self.fc1.weight.device == action.device == state_code.device):
            raise _capi.NdpError("Discriminator parameters are on %s, inputs on %s / %s"
                                 % (self.fc1.weight.device, action.device, state_code.device))
        return _DForward.apply(action, state_code, self, *self._param_list())


def collapse_batch(batch):
    """[A, B, ...] -> [A*B, ...] for 3-D and 5-D batches (models/gan.py:113-122)."""
    if batch.dim() in (3, 5):
        return batch.reshape((-1,) + tuple(batch.shape[2:]))
    print("Error: No need to collapse")
    return batch


def uncollapse_batch(batch, num_sample):
    """Inverse of collapse_batch for `num_sample` rows per group.  (The reference
    version, models/gan.py:125-134, reads an undefined global `num_sample`; it is an
    explicit argument here.)"""
    if batch.dim() in (2, 4):
        return batch.reshape((batch.shape[0] // num_sample, num_sample) + tuple(batch.shape[1:]))
    print("Error: No need to un-collapse")
    return batch
