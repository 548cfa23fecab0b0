import sys, os
sys.path.insert(0, "/root/repo")
import torch
from oracle import gan_oracle as O
from ndivplanning_amd.models.gan import Decoder, Discriminator
from ndivplanning_amd.trainer import GanTrainer
DEV = "cuda:0"
batch, k, world = int(os.environ.get("B", 256)), 6, int(os.environ.get("W", 8))
codes, actions, noise = O.synthetic_batch(31, batch, k, steps=1)
noise = noise[0]
flat, per = codes.shape[0], codes.shape[0] // world
g, d = O.init_params(0, 2)
def flatp(p): return torch.cat([v.reshape(-1) for v in p.values()])
names = []
off = 0
for n_, v in g.items():
    names.append((n_, off, off + v.numel())); off += v.numel()
for r in range(world):
    sl = slice(r * per, (r + 1) * per)
    sm = O.StepMath({n: v.clone() for n, v in g.items()}, {n: v.clone() for n, v in d.items()})
    sm.g_forward(codes[sl], actions[sl], noise[sl])
    dg = sm.d_grads(inv_m=1.0 / (flat * k))
    gg = sm.g_grads(inv_m=1.0 / (flat * k))       # D not updated
    dec, dis = Decoder(2), Discriminator(); dec.load_state_dict(g); dis.load_state_dict(d)
    t = GanTrainer(dec.to(DEV), dis.to(DEV), flat=per, num_sample=k, flat_global=flat, use_graph=False, reduce_fn=lambda grad: None)
    t.codes.copy_(codes[sl]); t.actions.copy_(actions[sl]); t.noise.copy_(noise[sl])
    t._phase_a(True)
    t._phase_b()
    torch.cuda.synchronize()
    ed = (t.d_grad.cpu() - flatp(dg)).abs().max().item()
    eg = (t.g_grad.cpu() - flatp(gg)).abs()
    print("rank %d: D grad err %.3e (scale %.2e)  G grad err %.3e (scale %.2e)" % (r, ed, flatp(dg).abs().max(), eg.max().item(), flatp(gg).abs().max()))
    if r == 0:
        for n_, a, b in names:
            print("    %-12s err %.3e  scale %.3e" % (n_, eg[a:b].max().item(), flatp(gg)[a:b].abs().max().item()))
        ah = (t.action_hat[:per * k].cpu() - sm.out["action_hat"]).abs().max().item()
        print("    action_hat err %.3e; pair_div %.6f vs %.6f" % (ah, t.losses()[2], sm.out["pair_div"].item()))
