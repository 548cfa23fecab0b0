"""Diagnostic (GPU box): where the time of GanTrainer.step_many_from_host goes -- the pinned upload alone, the
device-side staging copy alone, the graph alone, and their combinations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ndivplanning_amd.models.gan import Decoder, Discriminator
from ndivplanning_amd.trainer import GanTrainer
from oracle import gan_oracle as O

dev = torch.device("cuda:0")
spl, batch, k = 16, 64, 6
flat = batch * 7
g, d = O.init_params(0, 2)
dec, dis = Decoder(2), Discriminator(); dec.load_state_dict(g); dis.load_state_dict(d)
tr = GanTrainer(dec.to(dev), dis.to(dev), flat=flat, num_sample=k, steps_per_launch=spl)
hc = torch.randn(spl, flat, 256).pin_memory(); ha = torch.rand(spl, flat, 4).pin_memory()
tr.codes_slots.copy_(hc); tr.actions_slots.copy_(ha)

def timeit(name, fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-46s %8.1f us per launch of %d steps = %.2f us/step" % (name, dt * 1e6, spl, dt * 1e6 / spl), flush=True)

stage_c, stage_a = torch.empty_like(tr.codes_slots), torch.empty_like(tr.actions_slots)
timeit("graph replay alone", lambda: tr.step_many())
timeit("H2D pinned -> device (7.5 MB), same stream", lambda: (stage_c.copy_(hc, non_blocking=True), stage_a.copy_(ha, non_blocking=True)))
timeit("D2D staging -> slots", lambda: (tr.codes_slots.copy_(stage_c, non_blocking=True), tr.actions_slots.copy_(stage_a, non_blocking=True)))
timeit("H2D straight into the slots + replay (one stream)", lambda: (tr.codes_slots.copy_(hc, non_blocking=True), tr.actions_slots.copy_(ha, non_blocking=True), tr.step_many()))
timeit("step_many_from_host (copy stream + staging)", lambda: tr.step_many_from_host(hc, ha))
# host cost of the call itself
t0 = time.perf_counter()
for _ in range(40): tr.step_many_from_host(hc, ha)
host = (time.perf_counter() - t0) / 40
torch.cuda.synchronize()
print("host time per step_many_from_host call: %.1f us" % (host * 1e6))
t0 = time.perf_counter()
for _ in range(40): tr.step_many()
host = (time.perf_counter() - t0) / 40
torch.cuda.synchronize()
print("host time per step_many call: %.1f us" % (host * 1e6))

# --- is the upload concurrent with the graph?  events on both streams, one launch
copy_stream = torch.cuda.Stream(dev)
e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
torch.cuda.synchronize()
e[0].record()                      # main: start
tr.step_many()                     # graph k on main
e[1].record()
with torch.cuda.stream(copy_stream):
    e[2].record(copy_stream)
    stage_c.copy_(hc, non_blocking=True); stage_a.copy_(ha, non_blocking=True)
    e[3].record(copy_stream)
torch.cuda.synchronize()
print("graph %.0f us | upload on the copy stream: starts %.0f us after the graph starts, takes %.0f us (alone: see above)"
      % (e[0].elapsed_time(e[1]) * 1e3, e[0].elapsed_time(e[2]) * 1e3, e[2].elapsed_time(e[3]) * 1e3))
# --- the bench's loop shape: rotating pool of 4 host batches, 125 launches
pool_c = [torch.randn(spl, flat, 256).pin_memory() for _ in range(4)]
pool_a = [torch.rand(spl, flat, 4).pin_memory() for _ in range(4)]
for n in (40, 125):
    for _ in range(5): tr.step_many_from_host(pool_c[0], pool_a[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): tr.step_many_from_host(pool_c[i % 4], pool_a[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("pool of 4, %3d launches: %.1f us per launch = %.2f us/step" % (n, dt * 1e6, dt * 1e6 / spl))
