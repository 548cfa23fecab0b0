"""Diagnostic (GPU box): time forward-model training steps.  N=<batch> STEPS=<k>."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ndivplanning_amd.forward_trainer import ForwardModelTrainer
from ndivplanning_amd.models import forward_encoder as FE
n, steps = int(os.environ.get("N", 8)), int(os.environ.get("STEPS", 20))
dev = "cuda:0"
torch.manual_seed(0)
model = FE.ForwardAutoencoder()
model.decoder.weight_init(0.0, 0.02); model.encoder.weight_init(0.0, 0.02)
model = model.to(dev).train()
tr = ForwardModelTrainer(model, batch=n)
gen = torch.Generator().manual_seed(1)
cur = (torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1).to(dev)
fut = (torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1).to(dev)
act = (torch.rand(n, 4, generator=gen) * 2 - 1).to(dev)
for _ in range(3):
    tr.step(cur, fut, act)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(cur, fut, act)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
macs = 1.73e9 * 3 * n
print("batch %d: %.3f ms/step, %.1f steps/s, %.1f images/s, ~%.1f TFLOP/s (fwd+bwd ~ %.1f GMAC)" % (n, dt * 1e3, 1 / dt, n / dt, 2 * macs / dt / 1e12, macs / 1e9))
print("loss", tr.loss.item())
from ndivplanning_amd import _capi
was = _capi.load().ndp_fm_side_stream(0)          # one stream: the per-launch durations do not overlap
_capi.timing_enable(True)
for _ in range(3):
    tr.step(cur, fut, act)
torch.cuda.synchronize()
timed = _capi.timing_collect()
_capi.timing_enable(False)
_capi.load().ndp_fm_side_stream(was)
tot = sum(v[0] for v in timed.values())
print("per label (HIP events, 3 steps, one stream): total %.3f ms/step, %d launches/step" % (tot / 3, sum(v[1] for v in timed.values()) // 3))
groups = {}
for name, (ms, cnt) in timed.items():
    g = groups.setdefault(name.split("[")[0], [0.0, 0]); g[0] += ms; g[1] += cnt
for name, (ms, cnt) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print("  = %-26s %3d launches/step  %8.1f us each  %7.1f us/step %5.1f%%" % (name, cnt // 3, ms / cnt * 1e3, ms / 3 * 1e3, 100 * ms / tot))
for name, (ms, cnt) in sorted(timed.items(), key=lambda kv: -kv[1][0]):
    print("  %-28s %3d launches/step  %8.1f us each  %7.1f us/step %5.1f%%" % (name, cnt // 3, ms / cnt * 1e3, ms / 3 * 1e3, 100 * ms / tot))
if os.environ.get("GRAPH"):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tr.step(cur, fut, act)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        tr.step(cur, fut, act)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("graph replay: batch %d: %.3f ms/step" % (n, dt * 1e3))
