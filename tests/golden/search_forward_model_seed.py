#!/usr/bin/env python3
"""Search (seed, data_seed) for tests/golden/make_golden_forward_model.py: a case whose FIRST iteration decides every ReLU
unambiguously -- at every ReLU site the smallest |pre-activation| (fp64) is at least 4 times the site's fp32 rounding
error (root mean square of fp32 - fp64 over the site) -- so that two correct fp32 implementations take the same branch everywhere,
their first-iteration gradients agree to rounding, Adam moves the parameters the same way, and the SECOND iteration can be
compared as tightly as the first.  (Among ~2 M pre-activations per image a handful is normally within rounding of zero;
about one seed in thirty has none.)  Uses the oracle's restatement only (test infrastructure); prints the candidates.

Usage: python tests/golden/search_forward_model_seed.py [first_seed] [count] [n_images]
"""
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import forward_model_oracle as FO  # noqa: E402


def preacts(state, cur, act):
    rec, sites, real = {}, iter(FO.RELU_SITES), F.relu

    def spy(x, *a, **kw):
        rec[next(sites)] = x.detach().clone()
        return real(x, *a, **kw)
    FO.F.relu = spy
    try:
        with torch.no_grad():
            FO.forward(state, cur, act, training=True)
    finally:
        FO.F.relu = real
    return rec


def margins(seed, data_seed, n):
    state = FO.init_forward_model_state(seed)
    gen = torch.Generator().manual_seed(data_seed)
    frames = torch.rand(n, 3, 3, 128, 128, generator=gen) * 2.0 - 1.0
    actions = torch.rand(n, 3, 4, generator=gen) * 2.0 - 1.0
    cur, act = frames[:, 0], actions[:, 0]
    s32 = {k: v.clone() for k, v in state.items()}
    s64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
    p32, p64 = preacts(s32, cur, act), preacts(s64, cur.double(), act.double())
    worst = float("inf")
    detail = {}
    for site in FO.RELU_SITES:
        err = float((p32[site].double() - p64[site]).pow(2).mean().sqrt())
        low = float(p64[site].abs().min())
        detail[site] = (low, err)
        worst = min(worst, low / max(err, 1e-30))
    return worst, detail


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    torch.set_num_threads(8)
    best = (0.0, None)
    for seed in range(first, first + count):
        w, detail = margins(seed, seed + 1000, n)
        if w > best[0]:
            best = (w, seed)
        flag = "  <-- candidate" if w >= 4.0 else ""
        print("seed %d data_seed %d: smallest |pre-activation| / rms fp32 error over the sites = %.2f%s" % (seed, seed + 1000, w, flag), flush=True)
        if w >= 6.0:
            break
    print("best", best)
    w, detail = margins(best[1], best[1] + 1000, n)
    for site, (low, err) in detail.items():
        print("  %-6s min |x| %.3e   rms fp32 error %.3e   ratio %.1f" % (site, low, err, low / err))


if __name__ == "__main__":
    main()
