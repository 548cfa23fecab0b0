#!/usr/bin/env python3
"""Golden vector for the image encoder, from the REFERENCE's own `models.image_autoencoder.Encoder`
(imports with torch alone).  Runs only in the build container (needs /root/reference); the .npz travels.

The 8.4 M weights are not stored: the fixture records the seeds (the reference module is constructed right after
torch.manual_seed(7), its three applied BatchNorms then get parameters / running statistics from
torch.Generator().manual_seed(8), images from seed 9 -- oracle.encoder_oracle.init_encoder_state /
synthetic_images restate exactly that), a checksum of the resulting state_dict so that a test can tell an RNG-stream
change from a wrong result, the codes of 3 images and per-layer checksums.

Usage: python tests/golden/make_golden_encoder.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    sys.path.insert(0, REF)
    from models.image_autoencoder import Encoder          # the reference's module
    from oracle import encoder_oracle as EO                # only for the BN perturbation recipe and the images
    torch.set_num_threads(1)
    torch.manual_seed(7)
    enc = Encoder()
    pert = EO.init_encoder_state(7, bn_seed=8)              # re-seeds; enc was built first from the same seed
    sd = enc.state_dict()
    for i in (1, 2, 3):
        for key in ("weight", "bias", "running_mean", "running_var"):
            sd["conv%d_bn.%s" % (i, key)].copy_(pert["conv%d_bn.%s" % (i, key)])
    enc.eval()
    x = EO.synthetic_images(9, 3)
    feats = {}
    with torch.no_grad():
        h = x
        import torch.nn.functional as F
        h = F.relu(enc.conv1_bn(enc.conv1(h))); feats["feat1"] = h
        h = F.relu(enc.conv2_bn(enc.conv2(h))); feats["feat2"] = h
        h = F.relu(enc.conv3_bn(enc.conv3(h))); feats["feat3"] = h
        h = F.relu(enc.conv4(h)); feats["feat4"] = h
        h = F.relu(enc.conv5(h)); feats["feat5"] = h
        codes = enc(x)                                       # the reference's forward, image_autoencoder.py:35-49
        assert torch.equal(codes, enc.conv6(h))
    rec = {"codes": codes.reshape(3, 128).numpy(), "seeds": np.array([7, 8, 9])}
    for k, v in feats.items():
        v64 = v.double()
        rec[k + "_sums"] = np.array([v64.sum().item(), v64.abs().sum().item()])
    w64 = torch.cat([v.double().reshape(-1) for k, v in sorted(enc.state_dict().items()) if v.dtype.is_floating_point])
    rec["state_checksum"] = np.array([w64.sum().item(), w64.abs().sum().item(), float(w64.numel())])
    path = os.path.join(HERE, "encoder_case.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path), "bytes; codes abs max", float(codes.abs().max()))


if __name__ == "__main__":
    main()
