#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the .npz files it
writes are committed and travel to the GPU box, the reference does not.

How the reference is driven:
* `models.gan` and `diversity` import third-party modules they never use
  (torchvision, imageio, spectral_normalization).  Those are absent here, so
  empty stub modules are registered for them before the import; no reference
  code path exercised below touches them.
* `train_gan.py` cannot be imported (h5py, dotmap, visdom, cv2 absent; no
  data; yaml.load signature), so its loop body (train_gan.py:126-203) is
  driven here line by line on top of the reference's own `Decoder`,
  `Discriminator`, `diversity.compute_pairwise_divergence`,
  `nn.BCEWithLogitsLoss` and `optim.Adam` objects.
* Inputs, noise and initial weights are stored explicitly, so nothing depends
  on RNG-stream equivalence.

Usage: python tests/golden/make_golden.py   (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    for name in ("torchvision", "torchvision.models", "torchvision.datasets",
                 "torchvision.transforms", "imageio", "spectral_normalization"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tv.models = sys.modules["torchvision.models"]
    tv.datasets = sys.modules["torchvision.datasets"]
    tv.transforms = sys.modules["torchvision.transforms"]
    sys.modules["spectral_normalization"].SpectralNorm = object
    sys.path.insert(0, REF)
    import diversity as ref_div                      # noqa: E402
    from models.gan import Decoder, Discriminator    # noqa: E402
    return ref_div, Decoder, Discriminator


def _np(t):
    return t.detach().cpu().numpy().copy()


def run_case(ref_div, Decoder, Discriminator, *, seed, batch, num_sample, noise_dim,
             steps, dsteps, factor, keep, traj_len=8, lr=2e-4, coincide=False):
    """Drive `steps` iterations of train_gan.py:126-203 in codes mode and record I/O.
    keep: 'full' stores every intermediate at every step; 'final' stores losses per
    step, step-0 activations and the final parameters."""
    torch.manual_seed(seed)                                           # train_gan.py:65
    decoder = Decoder(noise_dim=noise_dim)                            # train_gan.py:89
    discriminator = Discriminator()                                   # train_gan.py:90
    decoder.weight_init(mean=0.0, std=0.02)                           # no-op on Linear (gan.py:15-18)
    discriminator.weight_init(mean=0.0, std=0.02)
    g_opt = torch.optim.Adam([{"params": decoder.parameters()}], lr=lr, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(discriminator.parameters(), lr=lr, betas=(0.5, 0.999))

    gen = torch.Generator().manual_seed(seed + 1000)
    flat = batch * (traj_len - 1)
    m_rows = flat * num_sample
    codes = torch.randn(flat, 256, generator=gen)
    actions = torch.rand(flat, 4, generator=gen) * 2.0 - 1.0
    noise_all = torch.rand(steps, flat, num_sample, noise_dim, generator=gen)
    if coincide:
        # two identical noise samples in every row -> d_ij == 0 off the diagonal for z and x
        noise_all[:, :, 1, :] = noise_all[:, :, 0, :]

    rec = {"codes": _np(codes), "actions": _np(actions), "noise": _np(noise_all),
           "meta": np.array([seed, batch, num_sample, noise_dim, steps, dsteps, traj_len], dtype=np.int64),
           "factor": np.array(factor, dtype=np.float64), "lr": np.array(lr, dtype=np.float64)}
    for n, p in decoder.state_dict().items():
        rec["g0." + n] = _np(p)
    for n, p in discriminator.state_dict().items():
        rec["d0." + n] = _np(p)

    bce = torch.nn.BCEWithLogitsLoss
    for s in range(steps):
        noises = noise_all[s]
        # train_gan.py:140, 156
        action_unsqueeze = torch.repeat_interleave(actions, repeats=num_sample, dim=0)
        codes_unsqueeze = torch.repeat_interleave(codes, repeats=num_sample, dim=0)
        # train_gan.py:42-47, 159-165
        code_e = codes[:, None, :].expand(-1, num_sample, -1)
        diverse_codes = torch.cat([code_e, noises], dim=2)[..., None, None]
        noises5 = noises[..., None, None]
        action_hat = decoder(diverse_codes.view(-1, diverse_codes.size(2)))
        full = keep == "full" or s == 0
        if full:
            rec["s%d.action_hat" % s] = _np(action_hat)
        for it in range(dsteps):                                      # train_gan.py:172-184
            l_real = discriminator(action_unsqueeze, codes_unsqueeze)
            l_fake = discriminator(action_hat, codes_unsqueeze)
            D_loss = bce()(torch.squeeze(l_real), torch.ones(m_rows)) + \
                bce()(torch.squeeze(l_fake), torch.zeros(m_rows))
            d_opt.zero_grad()
            D_loss.backward(retain_graph=True)
            if full and it == dsteps - 1:
                rec["s%d.logits_real" % s] = _np(l_real)
                rec["s%d.logits_fake" % s] = _np(l_fake)
                if keep == "full":
                    for n, p in discriminator.named_parameters():
                        rec["s%d.dgrad.%s" % (s, n)] = _np(p.grad)
            d_opt.step()
        l_gen = discriminator(action_hat, codes_unsqueeze)            # train_gan.py:187-190
        G_loss = bce()(torch.squeeze(l_gen), torch.ones(m_rows))
        pair_div = ref_div.compute_pairwise_divergence(               # train_gan.py:193-196
            action_hat.view(flat, num_sample, -1), noises5.squeeze(3).squeeze(3))
        total = G_loss + factor * pair_div
        g_opt.zero_grad()
        total.backward()
        if full:
            rec["s%d.logits_gen" % s] = _np(l_gen)
            if keep == "full":
                for n, p in decoder.named_parameters():
                    rec["s%d.ggrad.%s" % (s, n)] = _np(p.grad)
        g_opt.step()
        rec["s%d.losses" % s] = np.array([D_loss.item(), G_loss.item(), pair_div.item()], dtype=np.float64)
        if keep == "full" and s in (0, steps - 1) or s == steps - 1:
            for n, p in decoder.state_dict().items():
                rec["s%d.g.%s" % (s, n)] = _np(p)
            for n, p in discriminator.state_dict().items():
                rec["s%d.d.%s" % (s, n)] = _np(p)
    return rec


def run_ndiv_cases(ref_div):
    """Stand-alone diversity.py vectors: loss + gradient, including the edge cases
    (coincident samples, K=2, K=1 -> 0/0 = NaN, trailing singleton dims)."""
    rec = {}
    gen = torch.Generator().manual_seed(7)
    cases = {"k6": (5, 6, 4, 2), "k32": (3, 32, 4, 2), "k2": (4, 2, 4, 2), "k3c5": (2, 3, 5, 3)}
    for name, (n, k, cx, cz) in cases.items():
        x = (torch.randn(n, k, cx, generator=gen)).requires_grad_(True)
        z = torch.rand(n, k, cz, generator=gen)
        loss = ref_div.compute_pairwise_divergence(x, z)
        loss.backward()
        rec[name + ".x"], rec[name + ".z"] = _np(x), _np(z)
        rec[name + ".loss"], rec[name + ".grad"] = _np(loss), _np(x.grad)
        rec[name + ".pair_x"] = _np(ref_div.compute_pair_distance(x.detach()))
        rec[name + ".pairwise_x"] = _np(ref_div.compute_pairwise(x.detach()))
        rec[name + ".unnormal_x"] = _np(ref_div.compute_pair_unnormal_distance(x.detach()))
    # coincident samples: x_1 == x_0 and z_2 == z_0 in every row
    x = torch.randn(4, 6, 4, generator=gen)
    x[:, 1] = x[:, 0]
    x.requires_grad_(True)
    z = torch.rand(4, 6, 2, generator=gen)
    z[:, 2] = z[:, 0]
    loss = ref_div.compute_pairwise_divergence(x, z)
    loss.backward()
    rec["coin.x"], rec["coin.z"], rec["coin.loss"], rec["coin.grad"] = _np(x), _np(z), _np(loss), _np(x.grad)
    # K = 1: every row sum is 0 -> NaN loss
    x = torch.randn(3, 1, 4, generator=gen).requires_grad_(True)
    z = torch.rand(3, 1, 2, generator=gen)
    loss = ref_div.compute_pairwise_divergence(x, z)
    loss.backward()
    rec["k1.x"], rec["k1.z"], rec["k1.loss"], rec["k1.grad"] = _np(x), _np(z), _np(loss), _np(x.grad)
    # trailing singleton dims as the caller passes them (train_gan.py:195)
    x = torch.randn(3, 6, 4, generator=gen)
    z = torch.rand(3, 6, 2, 1, 1, generator=gen)
    rec["sq.x"], rec["sq.z"] = _np(x), _np(z)
    rec["sq.loss"] = _np(ref_div.compute_pairwise_divergence(x, z.squeeze(3).squeeze(3)))
    return rec


def main():
    ref_div, Decoder, Discriminator = _import_reference()
    torch.set_num_threads(1)   # bit-stable fixture generation
    common = dict(ref_div=ref_div, Decoder=Decoder, Discriminator=Discriminator)
    cases = {
        # every intermediate, tiny shape: B=2, K=3 -> FLAT=14, M=42 (ragged vs 16/32-row tiles)
        "step_tiny_full": dict(seed=0, batch=2, num_sample=3, noise_dim=2, steps=3, dsteps=1, factor=0.1, keep="full"),
        # BASELINE config 1 shape: B=16, K=6 -> FLAT=112, M=672
        "step_cfg1": dict(seed=0, batch=16, num_sample=6, noise_dim=2, steps=3, dsteps=1, factor=0.1, keep="final"),
        # two D steps per G step, wider noise, coincident noise samples
        "step_dsteps2_nz5": dict(seed=3, batch=3, num_sample=5, noise_dim=5, steps=2, dsteps=2, factor=0.25,
                                 keep="final", coincide=True),
        # K=32 stress shape (config 5 per-row geometry), one trajectory batch of 4
        "step_k32": dict(seed=5, batch=4, num_sample=32, noise_dim=2, steps=2, dsteps=1, factor=0.1, keep="final"),
    }
    for name, kw in cases.items():
        rec = run_case(**common, **kw)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **rec)
        print("wrote %s (%d arrays, %.1f KB)" % (path, len(rec), os.path.getsize(path) / 1024))
    # scalars only at BASELINE config 2 shape (B=64, K=6)
    rec = run_case(**common, seed=0, batch=64, num_sample=6, noise_dim=2, steps=3, dsteps=1, factor=0.1, keep="final")
    slim = {k: v for k, v in rec.items() if k.endswith(".losses") or k in ("meta", "factor", "lr")}
    slim["s0.action_hat_sum"] = np.array([rec["s0.action_hat"].astype(np.float64).sum(),
                                          np.abs(rec["s0.action_hat"].astype(np.float64)).sum()])
    path = os.path.join(HERE, "step_cfg2_scalars.npz")
    np.savez_compressed(path, **slim)
    print("wrote %s" % path)
    rec = run_ndiv_cases(ref_div)
    path = os.path.join(HERE, "ndiv_cases.npz")
    np.savez_compressed(path, **rec)
    print("wrote %s (%d arrays, %.1f KB)" % (path, len(rec), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
