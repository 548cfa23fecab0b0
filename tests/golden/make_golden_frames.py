#!/usr/bin/env python3
"""Golden vector for the byte-frame input path (the device-side half of the input pipeline).

The reference's loader decodes every frame with PIL and turns it into the [-1, 1] float tensor the models see with
    (ToTensor()(Image.open(io.BytesIO(raw_bytes))) - 0.5) * 2.0            (utils/hdf5_load.py:9-11)
`utils/hdf5_load.py` cannot be imported here (h5py and torchvision are absent from this image and stay absent), so the
one torchvision function it uses is restated from its published definition -- ToTensor: uint8 HWC -> CHW,
`.to(float32).div(255)` -- and that single step is "parity unpinned" against torchvision itself.  Everything downstream IS
the reference: the codes stored here come from the reference's own `models.image_autoencoder.Encoder` run on those floats.

Stored: two JPEG files of synthetic 128x128 scenes (smooth shapes + texture: what a camera frame compresses like), the
bytes PIL decodes them to in this image, the 256-entry byte -> float table of the formula, and the reference encoder's
codes (module seeds as tests/golden/make_golden_encoder.py).  Runs only in the build container (needs /root/reference).

Usage: python tests/golden/make_golden_frames.py
"""
import io
import os
import sys

import numpy as np
import torch
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def scene(seed):
    """A synthetic 128x128 RGB frame: table-like gradient, a few discs and blocks, mild texture; every byte value occurs."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:128, 0:128].astype(np.float32)
    img = np.stack([40 + 1.2 * xx, 200 - 1.1 * yy, 90 + 0.5 * (xx + yy)], axis=2)
    for _ in range(6):
        cy, cx, r = rng.randint(10, 118), rng.randint(10, 118), rng.randint(5, 22)
        colour = rng.randint(0, 256, 3).astype(np.float32)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = colour
    y0, x0 = rng.randint(0, 90, 2)
    img[y0:y0 + 30, x0:x0 + 24] = rng.randint(0, 256, 3)
    img[:4, :, :] = np.linspace(0, 255, 128)[None, :, None]          # a ramp: all 256 byte values survive as neighbours
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def to_norm_tensor(frames_u8):
    """utils/hdf5_load.py:9-11 on decoded frames [n,128,128,3]: ToTensor (HWC bytes -> CHW float32 / 255), - 0.5, * 2."""
    t = torch.from_numpy(frames_u8).permute(0, 3, 1, 2).contiguous()
    return (t.to(torch.float32).div(255) - 0.5) * 2.0


def main():
    sys.path.insert(0, REF)
    from models.image_autoencoder import Encoder          # the reference's module
    from oracle import encoder_oracle as EO
    torch.set_num_threads(1)
    jpegs, frames = [], []
    for seed in (11, 12):
        buf = io.BytesIO()
        Image.fromarray(scene(seed)).save(buf, format="JPEG", quality=90)
        raw = buf.getvalue()
        jpegs.append(np.frombuffer(raw, dtype=np.uint8))
        frames.append(np.array(Image.open(io.BytesIO(raw)), dtype=np.uint8))
    frames = np.stack(frames)
    lut = ((torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255) - 0.5) * 2.0).numpy()
    x = to_norm_tensor(frames)
    assert np.array_equal(x.numpy(), lut[frames].transpose(0, 3, 1, 2))
    torch.manual_seed(7)
    enc = Encoder()
    pert = EO.init_encoder_state(7, bn_seed=8)
    sd = enc.state_dict()
    for i in (1, 2, 3):
        for key in ("weight", "bias", "running_mean", "running_var"):
            sd["conv%d_bn.%s" % (i, key)].copy_(pert["conv%d_bn.%s" % (i, key)])
    enc.eval()
    with torch.no_grad():
        codes = enc(x).reshape(2, 128).numpy()
    path = os.path.join(HERE, "frames_case.npz")
    np.savez_compressed(path, jpeg0=jpegs[0], jpeg1=jpegs[1], frames_u8=frames, lut=lut, codes=codes, seeds=np.array([7, 8]),
                        byte_values_present=np.array([len(np.unique(frames))]))
    print("wrote", path, os.path.getsize(path), "bytes;", len(np.unique(frames)), "distinct byte values; codes abs max",
          float(np.abs(codes).max()))


if __name__ == "__main__":
    main()
