"""CPU: the flat parameter vector of the forward-model kernels (include/ndp.h, ndp_fm_layout) against the module's own
tensors -- packing and unpacking are inverse, padded entries are zero, every float of the vector belongs to exactly one
tensor -- and the host-side refusals that need no GPU."""
import ctypes

import pytest
import torch

from ndivplanning_amd import _capi
from ndivplanning_amd.models import forward_encoder as FE
from oracle import forward_model_oracle as FO


@pytest.fixture(scope="module")
def lib():
    from ndivplanning_amd import _build
    _build.build()
    return _capi.load()


def test_layout_tiles_the_vector_and_round_trips(lib):
    total = lib.ndp_fm_param_floats()
    spans = []
    for what, count in ((0, 14), (1, 14), (2, 10), (3, 10)):
        for i in range(count):
            off, d = FE._layout(lib, what, i)
            spans.append((off, off + d[0] * d[1] * d[2]))
    spans.sort()
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))          # back to back, no overlap
    assert all(s[0] % 4 == 0 for s in spans)                             # float4 loads
    state = FO.init_forward_model_state(1)
    model = FE.ForwardAutoencoder()
    model.load_state_dict(state)
    with torch.no_grad():                                                # make every tensor distinguishable
        for i, bn in enumerate(FE.BN_NAMES):
            FE._module_tensor(model, bn).weight.add_(0.01 * i)
            FE._module_tensor(model, bn).running_var.add_(0.1 * i)
    params, stats = FE.pack_module(model, "cpu")
    back = FE.unpack_vector(params, model)
    sd = model.state_dict()
    assert sorted(back) == sorted(k for k in sd if "running_" not in k and "num_batches" not in k
                                  and not k.startswith(("encoder.conv4_bn", "encoder.conv5_bn")))
    assert all(torch.equal(v, sd[k]) for k, v in back.items())
    # the real entries account for every non-zero of the vector: the padding is zero
    assert int((params != 0).sum()) == sum(int((v != 0).sum()) for v in back.values())
    # reference layouts: Conv2d [co][ci][kh][kw] -> [co][kh][kw][ci pad], ConvTranspose2d [ci][co][kh][kw] -> [ci pad][kh][kw][co]
    off, d = FE._layout(lib, 0, 0)
    assert d[:3] == [64, 9, 32] and params[off + (5 * 9 + 3 * 1 + 2) * 32 + 1] == sd["encoder.conv1.weight"][5, 1, 1, 2]
    off, d = FE._layout(lib, 0, 6)
    assert d[:3] == [160, 16, 1024] and params[off + (7 * 16 + 4 * 2 + 3) * 1024 + 9] == sd["decoder.deconv1.weight"][7, 9, 2, 3]
    clone = FE.ForwardAutoencoder()
    FE.unpack_into_module(clone, params, stats, batches_tracked=3)
    assert torch.equal(clone.decoder.deconv6.weight, model.decoder.deconv6.weight)
    assert torch.equal(clone.decoder.conv_refine_1_bn.running_var, model.decoder.conv_refine_1_bn.running_var)
    assert int(clone.encoder.conv2_bn.num_batches_tracked) == 3


def test_host_side_refusals(lib):
    off, dims = ctypes.c_int64(), (ctypes.c_int64 * 6)()
    assert lib.ndp_fm_layout(0, 14, ctypes.byref(off), dims) == 1 and b"out of range" in lib.ndp_last_error()
    assert lib.ndp_fm_workspace_floats(0) == 0 and lib.ndp_fm_workspace_floats(8) > lib.ndp_fm_workspace_floats(1) > 0
    assert lib.ndp_fm_workspace_offset(8, 999) == -1
    assert lib.ndp_fm_forward(None, None, None, None, 1, 1, None, None, None) == 1
    model = FE.ForwardAutoencoder().eval()
    with pytest.raises(_capi.NdpError):                                  # no CPU path for the kernels' case
        model(torch.zeros(1, 3, 128, 128), torch.zeros(1, 4))
    from ndivplanning_amd.forward_trainer import ForwardModelTrainer
    with pytest.raises(_capi.NdpError):
        ForwardModelTrainer(model, batch=2)
    model.train()
    x, a = torch.rand(1, 3, 128, 128) * 2 - 1, torch.rand(1, 4)
    with pytest.raises(_capi.NdpError):                                  # training mode neither
        model(x, a)
    # the operator composition the mirror keeps for Encoder / Decoder called on their own is the oracle's arithmetic
    y = model._forward_torch(x, a)
    want = FO.forward({k: v.clone() for k, v in model.state_dict().items()}, x, a, training=True)
    assert y.requires_grad and float((y.detach() - want).abs().max()) <= 1e-6
