"""Frozen image encoder that produces the GAN's condition codes -- mirror of the reference's
`models/image_autoencoder.py::Encoder` (image_autoencoder.py:14-49): 5 x conv3x3 stride 2
(BatchNorm only after the first three; conv4_bn / conv5_bn exist as attributes but are not
applied) + a 4x4 conv to 128 channels; 3x128x128 -> 128x1x1.

On the GAN path it runs in eval mode ahead of the step and its output is detached (train_gan.py:75-76,
152-153): that case -- CUDA input, eval mode, input without grad -- goes through the gfx950
kernels of `csrc/ndp_encoder.inc` (`ndp_encoder_forward`: implicit-GEMM convolutions on the fp32
matrix pipe, BatchNorm folded into the weights).  Training the autoencoder itself
(train_autoencoder.py) is outside this repository's scope; a forward in training mode or with
gradients keeps PyTorch's operators so that the class still behaves like an nn.Module there.
The class also exists so that the reference's whole-module encoder pickles load."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def normal_init(m, mean, std):
    if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
        m.weight.data.normal_(mean, std)
        m.bias.data.zero_()


class Encoder(nn.Module):
    _CHANNELS = (3, 64, 128, 256, 512, 1024)

    def __init__(self, d=16):
        super().__init__()
        ch = self._CHANNELS
        for i in range(5):
            setattr(self, "conv%d" % (i + 1), nn.Conv2d(ch[i], ch[i + 1], 3, 2, 1))
            setattr(self, "conv%d_bn" % (i + 1), nn.BatchNorm2d(ch[i + 1]))
        self.conv6 = nn.Conv2d(ch[5], 128, 4, 1, 0)

    def weight_init(self, mean, std):
        for name in self._modules:
            normal_init(self._modules[name], mean, std)

    def forward(self, x):
        # Every caller in the reference detaches the codes at once (train_gan.py:152-153, control_evaluation.py:110-111,
        # mpc_eval.py:139-140) with grad mode on and the loaded parameters still requiring grad: an eval-mode forward
        # is therefore NOT differentiable with respect to the encoder's parameters here; only an input that itself
        # requires grad (or training mode) selects the PyTorch operators.
        if x.dtype == torch.uint8:
            # decoded camera frames [n,128,128,3] (bytes, HWC): the first convolution normalises them as it gathers
            # (utils/hdf5_load.py:9-11's formula) -- eval mode only, the GAN path's case
            if self.training:
                raise RuntimeError("byte frames are accepted by the eval-mode Encoder only (normalise them for training: "
                                   "(x / 255 - 0.5) * 2, NCHW)")
            if not x.is_cuda:
                from .. import _capi
                raise _capi.NdpError("frames are on %s: the eval-mode Encoder computes only on a ROCm GPU (no CPU fallback)"
                                     % x.device)
            return _encoder_forward_hip(self, x)
        if not self.training and not (torch.is_grad_enabled() and x.requires_grad):
            if not x.is_cuda:
                # the GAN path's case has no CPU or eager fallback, like Decoder / Discriminator
                from .. import _capi
                raise _capi.NdpError("images are on %s: the eval-mode Encoder computes only on a ROCm GPU "
                                     "(no CPU fallback)" % x.device)
            return _encoder_forward_hip(self, x)
        return self._forward_torch(x)

    def __getstate__(self):
        state = self.__dict__.copy()                 # whole-module pickles carry no kernel scratch
        state.pop("_ndp_packed", None)
        state.pop("_ndp_ws", None)
        return state

    def _forward_torch(self, x):
        for i in (1, 2, 3):
            x = F.relu(getattr(self, "conv%d_bn" % i)(getattr(self, "conv%d" % i)(x)))
        x = F.relu(self.conv4(x))
        x = F.relu(self.conv5(x))
        return self.conv6(x)


def pack_encoder_params(enc):
    """The flat parameter buffer `ndp_encoder_forward` reads (layout: include/ndp.h): eval-mode BatchNorm of
    conv1..conv3 folded into weights and biases, conv1 as [27][64], conv2..6 as [Cout][KH][KW][Cin]."""
    parts = []
    with torch.no_grad():
        for i in range(1, 7):
            conv = getattr(enc, "conv%d" % i)
            w, b = conv.weight.detach().float(), conv.bias.detach().float()
            if i <= 3:
                bn = getattr(enc, "conv%d_bn" % i)
                scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                w = w * scale.view(-1, 1, 1, 1)
                b = (b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
            if i == 1:
                w = w.permute(1, 2, 3, 0).reshape(27, 64)            # [ci][kh][kw][co]
            else:
                w = w.permute(0, 2, 3, 1)                            # [co][kh][kw][ci]
            parts += [w.reshape(-1), b.reshape(-1)]
        return torch.cat(parts).contiguous()


def _encoder_state_key(enc):
    ts = list(enc.parameters()) + list(enc.buffers())
    return tuple((t.data_ptr(), t._version) for t in ts)


def _encoder_forward_hip(enc, x):
    from .. import _capi
    lib = _capi.load()
    u8 = x.dtype == torch.uint8
    if u8:
        if x.dim() != 4 or tuple(x.shape[1:]) != (128, 128, 3):
            raise _capi.NdpError("Encoder expects byte frames [n,128,128,3], got %s" % (tuple(x.shape),))
    else:
        _capi.require_gpu_f32(x, "images")
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 128, 128):
            raise _capi.NdpError("Encoder expects images [n,3,128,128], got %s" % (tuple(x.shape),))
    key = _encoder_state_key(enc)
    cache = enc.__dict__.get("_ndp_packed")
    if cache is None or cache[0] != key or cache[1].device != x.device:
        packed = pack_encoder_params(enc).to(x.device)
        if packed.numel() != lib.ndp_encoder_param_floats():
            raise _capi.NdpError("encoder parameter count %d != %d" % (packed.numel(), lib.ndp_encoder_param_floats()))
        cache = (key, packed)
        enc.__dict__["_ndp_packed"] = cache
    x = x.contiguous()
    n = x.shape[0]
    codes = torch.empty(n, 128, device=x.device, dtype=torch.float32)
    if n == 0:
        return codes.view(0, 128, 1, 1)
    ws = enc.__dict__.get("_ndp_ws")
    need = lib.ndp_encoder_workspace_floats(n)
    if ws is None or ws.numel() < need or ws.device != x.device:
        ws = torch.empty(need, device=x.device, dtype=torch.float32)
        enc.__dict__["_ndp_ws"] = ws
    with torch.cuda.device(x.device):
        fn = lib.ndp_encoder_forward_u8 if u8 else lib.ndp_encoder_forward
        _capi.check(fn(_capi.ptr(cache[1]), _capi.ptr(x), n, _capi.ptr(codes), _capi.ptr(ws), _capi.stream_ptr()),
                    "ndp_encoder_forward_u8" if u8 else "ndp_encoder_forward")
    return codes.view(n, 128, 1, 1)
