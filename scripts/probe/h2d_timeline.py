"""Diagnostic (GPU box): when does the upload of launch k+1 execute relative to the graph of launch k?"""
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
for _ in range(4): tr.step_many_from_host(hc, ha)
torch.cuda.synchronize()
st = tr._stage
E = lambda: torch.cuda.Event(enable_timing=True)
n = 8
ev = [[E() for _ in range(4)] for _ in range(n)]      # graph start, graph end, copy start, copy end
main = torch.cuda.current_stream(dev)
t0 = E(); t0.record(main)
for i in range(n):
    sset = st["next"]; st["next"] = 1 - sset
    codes = tr.codes_slots if sset == 0 else tr._alt_slots[0]
    actions = tr.actions_slots if sset == 0 else tr._alt_slots[1]
    with torch.cuda.stream(st["stream"]):
        st["stream"].wait_event(st["read"][sset])
        ev[i][2].record(st["stream"])
        codes.copy_(hc, non_blocking=True); actions.copy_(ha, non_blocking=True)
        ev[i][3].record(st["stream"])
        st["uploaded"][sset].record(st["stream"])
    main.wait_event(st["uploaded"][sset])
    ev[i][0].record(main)
    tr._replay(True, spl, sset)
    ev[i][1].record(main)
    st["read"][sset].record(main)
torch.cuda.synchronize()
for i in range(n):
    print("launch %d: graph %7.0f .. %7.0f us (%.0f)   upload %7.0f .. %7.0f us (%.0f)" % (
        i, t0.elapsed_time(ev[i][0]) * 1e3, t0.elapsed_time(ev[i][1]) * 1e3, ev[i][0].elapsed_time(ev[i][1]) * 1e3,
        t0.elapsed_time(ev[i][2]) * 1e3, t0.elapsed_time(ev[i][3]) * 1e3, ev[i][2].elapsed_time(ev[i][3]) * 1e3))
