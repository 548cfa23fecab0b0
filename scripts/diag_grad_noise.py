"""Diagnostic (GPU box): gradient error of the HIP path vs an fp64 oracle, next to the
error of the fp32 torch-CPU reference arithmetic vs the same fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden, golden_params
from oracle import gan_oracle as O
from ndivplanning_amd.models.gan import Decoder, Discriminator
from ndivplanning_amd.trainer import GanTrainer

DEV = "cuda:0"
def flat(p): return torch.cat([v.reshape(-1) for v in p.values()])

for case in ["step_tiny_full", "step_cfg1", "step_dsteps2_nz5", "step_k32"]:
    rec = load_golden(case)
    seed, batch, k, nz, steps, dsteps, traj = [int(v) for v in rec["meta"]]
    factor, lr = float(rec["factor"]), float(rec["lr"])
    g, d = golden_params(rec, "g0."), golden_params(rec, "d0.")
    codes, actions = torch.from_numpy(rec["codes"]), torch.from_numpy(rec["actions"])
    noise = torch.from_numpy(rec["noise"])[0]
    # fp64 and fp32 hand-written oracle from identical state, D update replaced by the fp64 result
    out = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        sm = O.StepMath({n: v.to(dt).clone() for n, v in g.items()}, {n: v.to(dt).clone() for n, v in d.items()}, lr=lr, pairwise_div_factor=factor)
        sm.g_forward(codes.to(dt), actions.to(dt), noise.to(dt))
        dg = sm.d_grads()
        gg = sm.g_grads()     # D NOT updated: isolates gradient noise from Adam
        out[name] = (flat(dg).double(), flat(gg).double(), sm.out["action_hat"].double())
    dec, dis = Decoder(nz), Discriminator()
    dec.load_state_dict(g); dis.load_state_dict(d)
    dec, dis = dec.to(DEV), dis.to(DEV)
    seen = []
    tr = GanTrainer(dec, dis, flat=codes.shape[0], num_sample=k, lr=lr, pairwise_div_factor=factor, use_graph=False,
                    reduce_fn=lambda gr: seen.append(gr.clone()))
    tr.codes.copy_(codes); tr.actions.copy_(actions); tr.noise.copy_(noise)
    tr._phase_a(True); dgrad = tr.d_grad.clone().cpu().double()
    tr._phase_b(); ggrad = tr.g_grad.clone().cpu().double()
    ah = tr.action_hat[:codes.shape[0]*k].cpu().double()
    d64, g64, a64 = out["f64"]; d32, g32, a32 = out["f32"]
    print("%-18s Dscale %.2e  D err hip %.2e  torch32 %.2e | Gscale %.2e  G err hip %.2e  torch32 %.2e | a_hat err hip %.2e torch32 %.2e" % (
        case, d64.abs().max(), (dgrad-d64).abs().max(), (d32-d64).abs().max(),
        g64.abs().max(), (ggrad-g64).abs().max(), (g32-g64).abs().max(),
        (ah-a64).abs().max(), (a32-a64).abs().max()))
