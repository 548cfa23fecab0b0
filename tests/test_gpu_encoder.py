"""GPU parity of the hand-written image encoder (csrc/ndp_encoder.inc, `ndp_encoder_forward`) against the oracle
restatement of the reference's Encoder and against the reference's own golden codes."""
import numpy as np
import pytest
import torch

from oracle import encoder_oracle as EO
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _encoder(state):
    from ndivplanning_amd.models.image_autoencoder import Encoder
    enc = Encoder()
    enc.load_state_dict(state, strict=False)
    return enc.to(DEV).eval()


def test_reference_golden_codes():
    g = load_golden("encoder_case")
    seed, bn_seed, img_seed = (int(v) for v in g["seeds"])
    enc = _encoder(EO.init_encoder_state(seed, bn_seed=bn_seed))
    with torch.no_grad():
        codes = enc(EO.synthetic_images(img_seed, 3).to(DEV))
    assert codes.shape == (3, 128, 1, 1)
    scale = np.abs(g["codes"]).max()
    assert np.abs(codes.reshape(3, 128).cpu().numpy() - g["codes"]).max() <= 1e-4 * scale      # fp32 tolerance, relative


@pytest.mark.parametrize("n", [1, 17, 130])
def test_matches_oracle_fp64_adjudicated(n):
    """Ragged M (n * OH * OW not a multiple of the 128-row tile), several row tiles per layer, split-K conv6."""
    state = EO.init_encoder_state(11, bn_seed=12)
    enc = _encoder(state)
    x = EO.synthetic_images(13 + n, n)
    with torch.no_grad():
        got = enc(x.to(DEV)).reshape(n, 128).cpu().double()
    ref32 = EO.encoder_forward(state, x).reshape(n, 128).double()
    ref64 = EO.encoder_forward(state, x, dtype=torch.float64).reshape(n, 128)
    scale = ref64.abs().max()
    bound = torch.maximum(1e-5 * scale * torch.ones_like(ref64), 4.0 * (ref32 - ref64).abs())
    assert bool(((got - ref64).abs() <= bound).all()), float(((got - ref64).abs() / scale).max())


def test_eval_no_grad_cuda_is_the_hip_path_and_everything_else_is_not():
    from ndivplanning_amd import _capi
    enc = _encoder(EO.init_encoder_state(1))
    x = EO.synthetic_images(2, 2).to(DEV)
    _capi.timing_enable(True)
    with torch.no_grad():
        enc(x)
    torch.cuda.synchronize()
    names = set(_capi.timing_collect())
    assert {"k_enc_conv1", "k_conv_gemm[2]", "k_conv_gemm[6]", "k_splitk_reduce"} <= names
    with torch.no_grad():
        enc(x)                                    # still enabled: launches are recorded
    assert _capi.timing_collect()
    out = enc(x)                                  # the reference's pattern: grad mode on, output detached by the caller
    assert _capi.timing_collect() and not out.requires_grad
    enc(x.clone().requires_grad_(True))           # an input that wants a gradient: PyTorch operators
    assert not _capi.timing_collect()
    enc.train()
    with torch.no_grad():
        enc(x)                                    # training mode: PyTorch operators (batch statistics), no HIP launch
    assert not _capi.timing_collect()
    _capi.timing_enable(False)
    with pytest.raises(_capi.NdpError):
        with torch.no_grad():
            enc.eval()(torch.zeros(1, 3, 64, 64, device=DEV))


def test_parameter_change_is_picked_up():
    enc = _encoder(EO.init_encoder_state(1))
    x = EO.synthetic_images(2, 2).to(DEV)
    with torch.no_grad():
        a = enc(x).clone()
        enc.conv6.bias.add_(1.0)
        b = enc(x)
    assert torch.allclose(b, a + 1.0, atol=1e-6)
