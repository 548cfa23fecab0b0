"""Action-conditioned next-frame model -- mirror of the reference's `models/forward_encoder.py`
(forward_encoder.py:20-114): `Encoder` (5 stride-2 3x3 convolutions, BatchNorm after the first three, a 4x4
convolution to the 128-d code; returns the code and the five feature maps), `Decoder` (a U-Net of six transposed
convolutions on cat([up, skip]) + BatchNorm + ReLU, two refinement convolutions, tanh) and `ForwardAutoencoder`
(z = cat([code, action]); training mode returns the predicted residual, eval mode state_cur + residual).

Same class names, constructor order (so that `torch.manual_seed(s); ForwardAutoencoder()` draws the same
parameters), attribute names and state_dict keys as the reference, so that its whole-module pickles load with
`ndivplanning_amd.train_forward_model.bind_reference_class_paths()`.

Where the arithmetic runs: every `ForwardAutoencoder.forward` on a ROCm GPU goes through `ndp_fm_forward`
(csrc/ndp_forward_model.inc: implicit-GEMM convolutions, transposed convolutions as four parity classes, BatchNorm as a
statistics + an elementwise pass) -- eval mode as control_evaluation.py / mpc_eval.py call it, training mode under
no_grad, and training mode WITH autograd recording, where the residual carries a grad_fn whose backward is
`ndp_fm_backward` (the reference's own loop, `loss = mse(model(cur, a), fut - cur); loss.backward(); optimizer.step()`,
train_forward_model.py:102-110, runs unchanged, with torch's optimizer on the module's parameters).  The fast way to
train is `ndivplanning_amd.forward_trainer.ForwardModelTrainer`, which owns the flat parameter vector the kernels read
and runs forward, loss, backward and Adam in HIP without repacking.  There is no CPU path and no gradient with respect
to the images or actions (the reference never asks for one): both raise.  `Encoder` / `Decoder` called on their own
(nobody in the reference does) keep PyTorch's operators."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _capi

ENC_CHANNELS = (3, 64, 128, 256, 512, 1024)
LAYER_NAMES = ("encoder.conv1", "encoder.conv2", "encoder.conv3", "encoder.conv4", "encoder.conv5", "encoder.conv6",
               "decoder.deconv1", "decoder.deconv2", "decoder.deconv3", "decoder.deconv4", "decoder.deconv5",
               "decoder.deconv6", "decoder.conv_refine_1", "decoder.conv_refine_2")
BN_NAMES = ("encoder.conv1_bn", "encoder.conv2_bn", "encoder.conv3_bn", "decoder.deconv1_bn", "decoder.deconv2_bn",
            "decoder.deconv3_bn", "decoder.deconv4_bn", "decoder.deconv5_bn", "decoder.deconv6_bn",
            "decoder.conv_refine_1_bn")


def normal_init(m, mean, std):
    if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
        # the reference writes through `.data` (forward_encoder.py:13-16); the same draws through no_grad bump the
        # tensors' version counters, which the packed-parameter cache of ForwardAutoencoder is keyed on
        with torch.no_grad():
            m.weight.normal_(mean, std)
            m.bias.zero_()


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        ch = ENC_CHANNELS
        for i in range(5):
            setattr(self, "conv%d" % (i + 1), nn.Conv2d(ch[i], ch[i + 1], 3, 2, 1))
            setattr(self, "conv%d_bn" % (i + 1), nn.BatchNorm2d(ch[i + 1]))
        self.conv6 = nn.Conv2d(ch[5], 128, 4, 1, 0)

    def weight_init(self, mean, std):
        for name in self._modules:
            normal_init(self._modules[name], mean, std)

    def forward(self, x):
        feats = []
        for i in (1, 2, 3):
            x = F.relu(getattr(self, "conv%d_bn" % i)(getattr(self, "conv%d" % i)(x)))
            feats.append(x)
        for i in (4, 5):                                   # conv4_bn / conv5_bn are never applied (forward_encoder.py:51-54)
            x = F.relu(getattr(self, "conv%d" % i)(x))
            feats.append(x)
        code = self.conv6(x)
        return code.view(code.size(0), -1), feats


class Decoder(nn.Module):
    _DECONV = ((132, 1024), (2048, 512), (1024, 256), (512, 128), (256, 64), (128, 32))

    def __init__(self):
        super().__init__()
        for i, (cin, cout) in enumerate(self._DECONV):
            first = i == 0
            setattr(self, "deconv%d" % (i + 1), nn.ConvTranspose2d(cin, cout, 4, 1 if first else 2, 0 if first else 1))
            setattr(self, "deconv%d_bn" % (i + 1), nn.BatchNorm2d(cout))
        self.conv_refine_1 = nn.Conv2d(32, 16, 3, 1, 1)
        self.conv_refine_1_bn = nn.BatchNorm2d(16)
        self.conv_refine_2 = nn.Conv2d(16, 3, 3, 1, 1)

    def weight_init(self, mean, std):
        for name in self._modules:
            normal_init(self._modules[name], mean, std)

    def forward(self, z, feats):
        skips = [None] + list(feats[::-1])                 # deconv2 takes feat_5, ..., deconv6 takes feat_1
        up = z
        for i in range(6):
            if skips[i] is not None:
                up = torch.cat([up, skips[i]], dim=1)
            up = F.relu(getattr(self, "deconv%d_bn" % (i + 1))(getattr(self, "deconv%d" % (i + 1))(up)))
        up = F.relu(self.conv_refine_1_bn(self.conv_refine_1(up)))
        return torch.tanh(self.conv_refine_2(up))


# ---------------------------------------------------------------- flat parameter vector <-> module
def _layout(lib, what, index):
    off, dims = ctypes.c_int64(), (ctypes.c_int64 * 6)()
    _capi.check(lib.ndp_fm_layout(what, index, ctypes.byref(off), dims), "ndp_fm_layout")
    return off.value, list(dims)


def _module_tensor(model, dotted):
    obj = model
    for part in dotted.split("."):
        obj = getattr(obj, part)
    return obj


def to_kernel_layout(weight, rows, cols):
    """Conv2d [co][ci][kh][kw] / ConvTranspose2d [ci][co][kh][kw] -> [dim0 padded to rows][kh][kw][dim1 padded to cols]
    (include/ndp.h)."""
    w = weight.detach().float().permute(0, 2, 3, 1)
    out = torch.zeros(rows, w.shape[1], w.shape[2], cols, dtype=torch.float32, device=w.device)
    out[: w.shape[0], :, :, : w.shape[3]] = w
    return out


def from_kernel_layout(flat_w, rows, taps, cols, shape):
    """Inverse of to_kernel_layout: back to the module's weight shape."""
    k = int(round(taps ** 0.5))
    w = flat_w.view(rows, k, k, cols)[: shape[0], :, :, : shape[1]]
    return w.permute(0, 3, 1, 2).contiguous()


def pack_module(model, device=None):
    """(params, running_stats): the flat vectors the kernels read, from a ForwardAutoencoder (layout: include/ndp.h)."""
    lib = _capi.load()
    device = device if device is not None else next(model.parameters()).device
    params = torch.zeros(lib.ndp_fm_param_floats(), dtype=torch.float32, device=device)
    stats = torch.zeros(lib.ndp_fm_stat_floats(), dtype=torch.float32, device=device)
    with torch.no_grad():
        for i, name in enumerate(LAYER_NAMES):
            mod = _module_tensor(model, name)
            off, d = _layout(lib, 0, i)
            n = d[0] * d[1] * d[2]
            params[off:off + n] = to_kernel_layout(mod.weight, d[0], d[2]).to(device).reshape(-1)
            boff, bd = _layout(lib, 1, i)
            params[boff:boff + mod.bias.numel()] = mod.bias.detach().float().to(device)
        for i, name in enumerate(BN_NAMES):
            bn = _module_tensor(model, name)
            c = bn.weight.numel()
            params[_layout(lib, 2, i)[0]:][:c] = bn.weight.detach().float().to(device)
            params[_layout(lib, 3, i)[0]:][:c] = bn.bias.detach().float().to(device)
            stats[_layout(lib, 4, i)[0]:][:c] = bn.running_mean.detach().float().to(device)
            stats[_layout(lib, 5, i)[0]:][:c] = bn.running_var.detach().float().to(device)
    return params, stats


def unpack_vector(vec, model=None):
    """name -> tensor in the module's own shapes, from a flat vector in the parameters' layout (parameters, gradients or
    Adam moments)."""
    lib = _capi.load()
    out = {}
    shapes = {}
    ref = model if model is not None else ForwardAutoencoder()
    for name in LAYER_NAMES:
        shapes[name] = tuple(_module_tensor(ref, name).weight.shape)
    for i, name in enumerate(LAYER_NAMES):
        off, d = _layout(lib, 0, i)
        n = d[0] * d[1] * d[2]
        out[name + ".weight"] = from_kernel_layout(vec[off:off + n], d[0], d[1], d[2], shapes[name])
        boff, _ = _layout(lib, 1, i)
        out[name + ".bias"] = vec[boff:boff + d[5]].clone()
    for i, name in enumerate(BN_NAMES):
        off, d = _layout(lib, 2, i)
        out[name + ".weight"] = vec[off:off + d[0]].clone()
        off, d = _layout(lib, 3, i)
        out[name + ".bias"] = vec[off:off + d[0]].clone()
    return out


def unpack_into_module(model, params, stats=None, batches_tracked=None):
    """Write the flat vectors back into the module's parameters and buffers (after HIP training)."""
    lib = _capi.load()
    tensors = unpack_vector(params, model)
    with torch.no_grad():
        for key, value in tensors.items():
            _module_tensor(model, key).copy_(value)
        if stats is not None:
            for i, name in enumerate(BN_NAMES):
                bn = _module_tensor(model, name)
                c = bn.weight.numel()
                bn.running_mean.copy_(stats[_layout(lib, 4, i)[0]:][:c])
                bn.running_var.copy_(stats[_layout(lib, 5, i)[0]:][:c])
                if batches_tracked is not None:
                    bn.num_batches_tracked.fill_(int(batches_tracked))


class _TrainingForward(torch.autograd.Function):
    """residual = model(state_cur, actions) in training mode, differentiable with respect to the model's parameters:
    forward = ndp_fm_forward (batch-statistics BatchNorm, running statistics moved), backward = ndp_fm_backward.
    The parameters are passed as inputs so that autograd routes their gradients; their order is `names`."""

    @staticmethod
    def forward(ctx, model, state_cur, actions, names, *params):
        out = model._forward_hip(state_cur, actions)
        ctx.model, ctx.names, ctx.n = model, names, int(state_cur.shape[0])
        ctx.serial = model.__dict__["_ndp_cache"]["serial"]
        return out

    @staticmethod
    def backward(ctx, d_resid):
        model = ctx.model
        lib = _capi.load()
        cache = model.__dict__.get("_ndp_cache")
        if cache is None or cache.get("serial") != ctx.serial or cache.get("consumed"):
            raise _capi.NdpError("backward through a ForwardAutoencoder forward whose activations are gone: another forward "
                                 "of the same module ran in between (one forward, one backward, as the reference's loop)")
        d = d_resid.detach().contiguous().float()
        grad = torch.empty_like(cache["params"])
        with _capi.on_device(d):
            _capi.check(lib.ndp_fm_backward(_capi.ptr(cache["params"]), _capi.ptr(d), ctx.n, _capi.ptr(grad), _capi.ptr(cache["ws"]),
                                            _capi.stream_ptr(d.device)), "ndp_fm_backward")
        cache["consumed"] = True
        by_name = unpack_vector(grad, model)
        return (None, None, None, None) + tuple(by_name.get(n) for n in ctx.names)


class ForwardAutoencoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.decoder = Decoder()

    def __getstate__(self):
        state = self.__dict__.copy()                 # whole-module pickles carry no kernel scratch
        for key in ("_ndp_cache",):
            state.pop(key, None)
        return state

    def _forward_torch(self, state_cur, actions):
        code, feats = self.encoder(state_cur)
        z = torch.cat([code, actions], dim=1).unsqueeze(2).unsqueeze(3)
        resid = self.decoder(z, feats)
        return resid if self.training else state_cur + resid

    def _versions(self):
        # (storage address, version) per tensor: an in-place write through torch bumps the version, a re-assigned or
        # moved parameter changes the address.  Writes through `.data` bump nothing -- call invalidate_cache() after one.
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def invalidate_cache(self):
        """Drop the packed parameter copies: the next forward re-reads the module's tensors."""
        self.__dict__.pop("_ndp_cache", None)

    def _forward_hip(self, state_cur, actions):
        lib = _capi.load()
        dev = state_cur.device
        n = int(state_cur.shape[0])
        cache = self.__dict__.get("_ndp_cache")
        if cache is None or cache["versions"] != self._versions() or cache["device"] != dev:
            params, stats = pack_module(self, dev)
            cache = {"versions": self._versions(), "device": dev, "params": params, "stats": stats, "ws": None, "n": 0,
                     "packed": False}
            self.__dict__["_ndp_cache"] = cache
        if cache["ws"] is None or cache["n"] < n:
            cache["ws"] = torch.empty(lib.ndp_fm_workspace_floats(n), dtype=torch.float32, device=dev)
            cache["n"], cache["packed"] = n, False
        u8 = state_cur.dtype == torch.uint8                 # decoded frames [n,128,128,3]: normalised by the kernels
        if u8 and tuple(state_cur.shape[1:]) != (128, 128, 3):
            raise _capi.NdpError("byte frames must be [n,128,128,3], got %s" % (tuple(state_cur.shape),))
        x = state_cur.detach().contiguous() if u8 else state_cur.detach().contiguous().float()
        a = actions.detach().contiguous().float()
        out = torch.empty(n, 3, 128, 128, dtype=torch.float32, device=dev)
        cache["serial"], cache["consumed"] = cache.get("serial", 0) + 1, False     # which forward the workspace holds
        with _capi.on_device(x):
            st = _capi.stream_ptr(dev)
            if not cache["packed"]:
                _capi.check(lib.ndp_fm_pack_params(_capi.ptr(cache["params"]), _capi.ptr(cache["ws"]), st), "ndp_fm_pack_params")
                cache["packed"] = True
            fn = lib.ndp_fm_forward_u8 if u8 else lib.ndp_fm_forward
            _capi.check(fn(_capi.ptr(cache["params"]), _capi.ptr(cache["stats"]), _capi.ptr(x), _capi.ptr(a), n,
                           1 if self.training else 0, _capi.ptr(out), _capi.ptr(cache["ws"]), st),
                        "ndp_fm_forward_u8" if u8 else "ndp_fm_forward")
        if self.training:                                  # batch statistics moved the running ones: hand them to the module
            with torch.no_grad():
                for i, name in enumerate(BN_NAMES):
                    bn = _module_tensor(self, name)
                    c = bn.weight.numel()
                    bn.running_mean.copy_(cache["stats"][_layout(lib, 4, i)[0]:][:c])
                    bn.running_var.copy_(cache["stats"][_layout(lib, 5, i)[0]:][:c])
                    bn.num_batches_tracked += 1
            cache["versions"] = self._versions()
        return out

    def forward(self, state_cur, actions):
        if not state_cur.is_cuda:
            raise _capi.NdpError("state_cur is on %s: ForwardAutoencoder computes only on a ROCm GPU (no CPU fallback)"
                                 % state_cur.device)
        if actions.device != state_cur.device:
            raise _capi.NdpError("state_cur is on %s, actions on %s" % (state_cur.device, actions.device))
        if torch.is_grad_enabled() and (state_cur.requires_grad or actions.requires_grad):
            raise NotImplementedError("ForwardAutoencoder gives no gradient with respect to its inputs (the reference "
                                      "never asks for one)")
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        if torch.is_grad_enabled() and self.training and named:
            # the reference's own training loop: the residual gets a grad_fn whose backward is the HIP backward pass
            return _TrainingForward.apply(self, state_cur, actions, tuple(n for n, _ in named), *[p for _, p in named])
        return self._forward_hip(state_cur, actions)
