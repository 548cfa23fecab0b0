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


def test_config4_image_conditioned_step_at_batch_128():
    """BASELINE configs[3] at its size: B = 128 trajectories x 8 frames = 1,024 unique 3x128x128 images through
    `ndp_encoder_forward` (two passes of 512), codes assembled as train_gan.py:152-155 does (current frame || target
    frame), then ONE fused train step at FLAT = 896, M = 5,376 against the oracle on the same codes.
    Encoder: fp64-adjudicated against the oracle on a 128-image slice that touches every trajectory and every frame
    position of both passes; finite + same scale on the other 896.  Reference: train_gan.py:127-203,
    models/image_autoencoder.py:35-49."""
    from oracle import gan_oracle as O
    from ndivplanning_amd.models.gan import Decoder, Discriminator
    from ndivplanning_amd.train_gan import encode_batch
    from ndivplanning_amd.trainer import GanTrainer
    batch, traj, k = 128, 8, 6
    state = EO.init_encoder_state(21, bn_seed=22)
    enc = _encoder(state)
    x = EO.synthetic_images(23, batch * traj)
    with torch.no_grad():
        codes128 = enc(x.to(DEV)).reshape(batch * traj, 128)
    assert codes128.shape == (1024, 128) and bool(torch.isfinite(codes128).all())
    idx = torch.arange(batch) * traj + (torch.arange(batch) % traj)          # trajectory b, frame b % 8
    ref32 = EO.encoder_forward(state, x[idx]).reshape(batch, 128).double()
    ref64 = EO.encoder_forward(state, x[idx], dtype=torch.float64).reshape(batch, 128)
    got = codes128[idx.to(DEV)].cpu().double()
    scale = ref64.abs().max()
    bound = torch.maximum(1e-5 * scale * torch.ones_like(ref64), 4.0 * (ref32 - ref64).abs())
    assert bool(((got - ref64).abs() <= bound).all()), float(((got - ref64).abs() / scale).max())
    rest = torch.ones(batch * traj, dtype=torch.bool)
    rest[idx] = False
    r = codes128[rest.to(DEV)]
    assert 0.5 * got.abs().mean() <= r.abs().mean().item() <= 2.0 * got.abs().mean()   # same distribution of inputs

    codes = encode_batch(codes128.reshape(batch, traj, 128), None, traj)
    assert codes.shape == (batch * (traj - 1), 256)
    # row f = b * 7 + t holds [code(frame t of b) || code(frame 7 of b)]
    assert torch.equal(codes[5 * 7 + 3, :128], codes128[5 * 8 + 3]) and torch.equal(codes[5 * 7 + 3, 128:], codes128[5 * 8 + 7])
    gen = torch.Generator().manual_seed(24)
    actions = torch.rand(batch * (traj - 1), 4, generator=gen) * 2.0 - 1.0
    noise = torch.rand(batch * (traj - 1), k, 2, generator=gen)
    g, d = O.init_params(0, 2)
    sm = O.StepMath({n: v.clone() for n, v in g.items()}, {n: v.clone() for n, v in d.items()})
    ref = sm.step(codes.cpu(), actions, noise)
    dec, dis = Decoder(2), Discriminator()
    dec.load_state_dict(g)
    dis.load_state_dict(d)
    tr = GanTrainer(dec.to(DEV), dis.to(DEV), flat=codes.shape[0], num_sample=k)
    assert tr.m == 5376
    tr.step(codes, actions.to(DEV), noise.to(DEV))
    d_loss, g_loss, pd = tr.losses()
    assert abs(d_loss - ref["d_loss"].item()) <= 1e-4 and abs(g_loss - ref["g_loss"].item()) <= 1e-4
    assert abs(pd - ref["pair_div"].item()) <= 1e-4 * max(1.0, abs(ref["pair_div"].item()))
    assert (tr.action_hat[:tr.m].cpu() - ref["action_hat"]).abs().max().item() <= 1e-4
    g0 = torch.cat([v.reshape(-1) for v in g.values()]).to(DEV)
    assert 0 < (tr.g_flat - g0).abs().max().item() <= 2.5 * 2e-4            # one Adam step happened, inside its bound


def test_byte_frames_give_the_float_path_s_codes_bit_for_bit_and_the_reference_s():
    """ndp_encoder_forward_u8: decoded camera frames [n,128,128,3] as bytes, normalised by conv1 as it gathers
    (utils/hdf5_load.py:9-11's formula as a 256-entry table).  Against (a) the float path fed the tensor the reference's
    loader would have built from the same bytes -- bit-identical, the arithmetic after the table is the same -- and (b) the
    codes of the reference's own Encoder on that tensor (tests/golden/frames_case.npz)."""
    from ndivplanning_amd import _capi
    from ndivplanning_amd.models.image_autoencoder import Encoder
    g = load_golden("frames_case")
    seed, bn_seed = (int(v) for v in g["seeds"])
    enc = Encoder()
    enc.load_state_dict(EO.init_encoder_state(seed, bn_seed=bn_seed), strict=False)
    enc = enc.to(DEV).eval()
    frames = torch.from_numpy(g["frames_u8"])
    x = torch.from_numpy(g["lut"][g["frames_u8"]].transpose(0, 3, 1, 2).copy())
    _capi.timing_enable(True)
    with torch.no_grad():
        got_u8 = enc(frames.to(DEV)).reshape(2, 128)
        got_f32 = enc(x.to(DEV)).reshape(2, 128)
    torch.cuda.synchronize()
    timed = _capi.timing_collect()
    _capi.timing_enable(False)
    assert timed["k_enc_conv1"][1] == 2                                 # both went through the HIP kernels
    assert torch.equal(got_u8, got_f32)
    scale = np.abs(g["codes"]).max()
    assert np.abs(got_u8.cpu().numpy() - g["codes"]).max() <= 1e-4 * scale
    # a bigger batch of random bytes (every value, every border case of the 3x3 stride-2 gather)
    gen = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (37, 128, 128, 3), generator=gen, dtype=torch.uint8)
    x = torch.from_numpy(g["lut"])[frames.long()].permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        assert torch.equal(enc(frames.to(DEV)), enc(x.to(DEV)))
    with pytest.raises(_capi.NdpError):
        enc(frames[:, :, :, :2].contiguous().to(DEV))                    # not [n,128,128,3]
    with pytest.raises(_capi.NdpError):
        enc(frames)                                                     # CPU bytes: no fallback either
