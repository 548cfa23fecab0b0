"""Diagnostic (GPU box): per-kernel event timings of the step at a few shapes, for whatever build NDP_LIB_PATH names
(ablation builds: -DNDP_EXP_NOLOAD no weight traffic, -DNDP_EXP_NOMFMA no matrix ops, -DNDP_EXP_NOSTORE no activation
stores, -DNDP_NO_ROTATION).  Ablated builds compute garbage; only their timings mean anything."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ndivplanning_amd import _capi
from ndivplanning_amd.models.gan import Decoder, Discriminator
from ndivplanning_amd.trainer import GanTrainer
from oracle import gan_oracle as O
dev = "cuda:0"
tag = os.environ.get("TAG", "base")
for batch, k in ((64, 6), (1024, 6), (128, 32)):
    g, d = O.init_params(0, 2)
    dec, dis = Decoder(2), Discriminator(); dec.load_state_dict(g); dis.load_state_dict(d)
    codes, actions, noise = O.synthetic_batch(0, batch, k, steps=1)
    tr = GanTrainer(dec.to(dev), dis.to(dev), flat=codes.shape[0], num_sample=k, steps_per_launch=4)
    tr.codes_slots.copy_(codes.expand(4, -1, -1)); tr.actions_slots.copy_(actions.expand(4, -1, -1))
    n = 40 if batch == 64 else 12
    for _ in range(3): tr.step_many()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step_many()
    torch.cuda.synchronize(); step_us = (time.perf_counter() - t0) / (4 * n) * 1e6
    tr.use_graph = False
    _capi.timing_enable(True)
    for _ in range(10): tr.step()
    torch.cuda.synchronize()
    t = _capi.timing_collect(); _capi.timing_enable(False)
    print("%-10s B=%4d K=%2d step %8.2f us | " % (tag, batch, k, step_us) +
          "  ".join("%s %.1f" % (nm.replace("k_", ""), 1e3 * ms / cnt) for nm, (ms, cnt) in t.items()), flush=True)
    del tr
