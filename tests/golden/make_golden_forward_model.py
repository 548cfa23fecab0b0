#!/usr/bin/env python3
"""Golden vector for the forward (next-frame) model, SURVEY.md section 8 row f4, from the REFERENCE's own
`models.forward_encoder.ForwardAutoencoder` driven exactly as train_forward_model.py:68-112 drives it (weight_init of the
decoder then the encoder, nn.MSELoss, optim.Adam(lr, betas (0.5, 0.999)), loss on the residual, two iterations).
Runs only in the build container (needs /root/reference); the .npz travels.

The 33 M parameters are not stored: the fixture records the seeds, checksums of the initial state_dict (so that a test
can tell an RNG-stream change from a wrong result), the two losses, samples and checksums of the predicted residual, and
per-tensor gradient / post-step parameter checksums.  The reference module imports matplotlib, imageio and torchvision
without using them: empty stub modules stand in (no permission was needed or denied).

Usage: python tests/golden/make_golden_forward_model.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def sums(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def main():
    # "forward_model_case": the first case (seed 5); "forward_model_case_sharp": a case whose first iteration decides every
    # ReLU unambiguously (tests/golden/search_forward_model_seed.py: at every ReLU site the smallest |pre-activation| is
    # >= 15 x the site's rms fp32 rounding error), so that the SECOND iteration of two correct fp32 implementations can be
    # compared as tightly as the first
    for name, seed, data_seed, n in (("forward_model_case", 5, 6, 2), ("forward_model_case_sharp", 210, 1210, 1)):
        one_case(name, seed, data_seed, n)


def one_case(case_name, seed, data_seed, n):
    mpl = _stub("matplotlib")
    mpl.pyplot = _stub("matplotlib.pyplot")
    _stub("imageio")
    tv = _stub("torchvision")
    tv.datasets, tv.transforms = _stub("torchvision.datasets"), _stub("torchvision.transforms")
    sys.path.insert(0, REF)
    from models.forward_encoder import ForwardAutoencoder          # the reference's module
    from oracle import forward_model_oracle as FO                   # only for the input recipe
    torch.set_num_threads(1)
    lr = 2e-4
    torch.manual_seed(seed)
    model = ForwardAutoencoder()
    model.decoder.weight_init(mean=0.0, std=0.02)                   # train_forward_model.py:69-70
    model.encoder.weight_init(mean=0.0, std=0.02)
    model.train()
    mse = torch.nn.MSELoss()
    opt = torch.optim.Adam([{"params": model.decoder.parameters()}, {"params": model.encoder.parameters()}], lr=lr,
                           betas=(0.5, 0.999))
    gen = torch.Generator().manual_seed(data_seed)
    frames = torch.rand(n, 3, 3, 128, 128, generator=gen) * 2.0 - 1.0          # [n, 3 frames, 3, 128, 128]
    actions = torch.rand(n, 3, 4, generator=gen) * 2.0 - 1.0
    rec = {"meta": np.array([seed, data_seed, n]), "lr": np.array(lr)}
    w64 = torch.cat([v.double().reshape(-1) for v in model.state_dict().values() if v.dtype.is_floating_point])
    rec["state_checksum"] = np.array([w64.sum().item(), w64.abs().sum().item(), float(w64.numel())])
    names = [k for k, _ in list(model.decoder.named_parameters(prefix="decoder")) + list(model.encoder.named_parameters(prefix="encoder"))]
    for it in range(2):
        cur, fut = frames[:, it], frames[:, it + 1]
        resid = model(cur, actions[:, it])
        loss = mse(resid, fut - cur)
        opt.zero_grad()
        loss.backward()
        rec["s%d.loss" % it] = np.array(loss.item())
        rec["s%d.resid_sums" % it] = sums(resid)
        rec["s%d.resid_sample" % it] = resid.detach()[:, :, ::16, ::16].numpy().copy()
        params = dict(list(model.decoder.named_parameters(prefix="decoder")) + list(model.encoder.named_parameters(prefix="encoder")))
        rec["s%d.grad_sums" % it] = np.stack([sums(params[k].grad) if params[k].grad is not None else np.zeros(3) for k in names])
        opt.step()
        rec["s%d.param_sums" % it] = np.stack([sums(params[k]) for k in names])
    rec["names"] = np.array(names)
    sd = model.state_dict()
    rec["running_sums"] = np.stack([sums(sd[k]) for k in sd if "running_" in k])
    path = os.path.join(HERE, case_name + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path), "bytes; losses", rec["s0.loss"], rec["s1.loss"])


if __name__ == "__main__":
    main()
