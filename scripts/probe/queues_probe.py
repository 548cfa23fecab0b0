"""Diagnostic (GPU box): how many other streams a process may use before the forward model's two-stream step slows down.
EXTRA=<k> dummy streams are created and used first; N=<batch>."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ndivplanning_amd.forward_trainer import ForwardModelTrainer
from ndivplanning_amd.models import forward_encoder as FE
n, extra = int(os.environ.get("N", 8)), int(os.environ.get("EXTRA", 0))
dev = "cuda:0"
streams = []
for i in range(extra):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        torch.zeros(1024, device=dev).add_(1.0)
    streams.append(st)
torch.cuda.synchronize()
torch.manual_seed(0)
model = FE.ForwardAutoencoder().to(dev).train()
tr = ForwardModelTrainer(model, batch=n)
gen = torch.Generator().manual_seed(1)
cur = (torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1).to(dev)
fut = (torch.rand(n, 3, 128, 128, generator=gen) * 2 - 1).to(dev)
act = (torch.rand(n, 4, generator=gen) * 2 - 1).to(dev)
for _ in range(3):
    tr.step(cur, fut, act)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(cur, fut, act)
    if os.environ.get("KEEP_BUSY") and streams:                 # the other streams stay in use during the run
        for st in streams:
            with torch.cuda.stream(st):
                torch.zeros(1024, device=dev).add_(1.0)
torch.cuda.synchronize()
print("extra streams %d, side priority %s, GPU_MAX_HW_QUEUES %s: %.3f ms/step" % (
    extra, os.environ.get("NDP_FM_SIDE_PRIORITY", "low"), os.environ.get("GPU_MAX_HW_QUEUES", "default"), (time.perf_counter() - t0) / 20 * 1e3))
