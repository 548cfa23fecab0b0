"""CPU tests of the encoder path: the oracle against the reference's golden vector, and the host-side BatchNorm
folding / weight re-layout (`pack_encoder_params`) against the oracle."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import encoder_oracle as EO
from conftest import load_golden


def _state_checksum(state):
    w64 = torch.cat([v.double().reshape(-1) for k, v in sorted(state.items()) if v.dtype.is_floating_point])
    return np.array([w64.sum().item(), w64.abs().sum().item(), float(w64.numel())])


def test_oracle_reproduces_reference_encoder_codes():
    g = load_golden("encoder_case")
    seed, bn_seed, img_seed = (int(v) for v in g["seeds"])
    state = EO.init_encoder_state(seed, bn_seed=bn_seed)
    # the fixture stores seeds, not 8.4 M weights: first make sure the RNG stream gave the same weights
    np.testing.assert_allclose(_state_checksum(state), g["state_checksum"], rtol=1e-12)
    torch.set_num_threads(1)
    codes = EO.encoder_forward(state, EO.synthetic_images(img_seed, 3)).reshape(3, 128)
    scale = np.abs(g["codes"]).max()
    assert np.abs(codes.numpy() - g["codes"]).max() <= 1e-6 * scale


def test_folded_packed_parameters_give_the_same_codes():
    """pack_encoder_params: BatchNorm folded into conv1..3, conv1 as [27][64], the others [Cout][KH][KW][Cin]."""
    from ndivplanning_amd.models.image_autoencoder import Encoder, pack_encoder_params
    state = EO.init_encoder_state(3, bn_seed=4)
    enc = Encoder()
    missing = enc.load_state_dict(state, strict=False)       # conv4_bn / conv5_bn keep their defaults (never applied)
    assert not missing.unexpected_keys
    enc.eval()
    packed = pack_encoder_params(enc)
    x = EO.synthetic_images(5, 2)
    want = EO.encoder_forward(state, x)
    o = 0
    h = x
    for i, (cin, cout, k) in enumerate([(3, 64, 3), (64, 128, 3), (128, 256, 3), (256, 512, 3), (512, 1024, 3), (1024, 128, 4)]):
        nw = cout * k * k * cin
        w, b = packed[o:o + nw], packed[o + nw:o + nw + cout]
        o += nw + cout
        w = w.reshape(cin, k, k, cout).permute(3, 0, 1, 2) if i == 0 else w.reshape(cout, k, k, cin).permute(0, 3, 1, 2)
        h = F.conv2d(h, w.contiguous(), b, stride=2 if i < 5 else 1, padding=1 if i < 5 else 0)
        if i < 5:
            h = F.relu(h)
    assert o == packed.numel()
    assert (h - want).abs().max() <= 1e-5 * want.abs().max()
    # and the module's training-mode / autograd body (PyTorch operators) is the reference computation
    with torch.no_grad():
        got = enc._forward_torch(x)
        assert torch.equal(got, want) or (got - want).abs().max() <= 1e-7
    # the eval-mode forward is the HIP path only: a CPU tensor is rejected, never computed eagerly
    from ndivplanning_amd._capi import NdpError
    import pytest
    with pytest.raises(NdpError):
        enc(x)


def test_byte_frames_decode_and_normalise_as_the_reference_loader_does():
    """tests/golden/frames_case.npz: JPEG bytes -> PIL -> bytes HWC -> (ToTensor - 0.5) * 2 (utils/hdf5_load.py:9-11) ->
    the reference Encoder's codes.  Here: this image's PIL decodes the stored JPEGs to the stored bytes, the mirror's
    `norm_frame` and the 256-entry table both give the formula's floats, and the oracle reproduces the stored codes."""
    import io
    from PIL import Image
    from ndivplanning_amd.utils.trajectory_loader import norm_frame
    g = load_golden("frames_case")
    decoded = np.stack([np.array(Image.open(io.BytesIO(g[k].tobytes())), dtype=np.uint8) for k in ("jpeg0", "jpeg1")])
    assert decoded.shape == (2, 128, 128, 3) and np.array_equal(decoded, g["frames_u8"])
    assert int(g["byte_values_present"][0]) == 256                      # every byte value is exercised
    lut = ((torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255) - 0.5) * 2.0).numpy()
    assert np.array_equal(lut, g["lut"]) and lut[0] == -1.0 and lut[255] == 1.0
    x = torch.stack([norm_frame(Image.open(io.BytesIO(g[k].tobytes()))) for k in ("jpeg0", "jpeg1")])
    assert x.shape == (2, 3, 128, 128) and np.array_equal(x.numpy(), g["lut"][g["frames_u8"]].transpose(0, 3, 1, 2))
    seed, bn_seed = (int(v) for v in g["seeds"])
    torch.set_num_threads(1)
    codes = EO.encoder_forward(EO.init_encoder_state(seed, bn_seed=bn_seed), x).reshape(2, 128).numpy()
    assert np.abs(codes - g["codes"]).max() <= 1e-6 * np.abs(g["codes"]).max()
