"""Diagnostic (GPU box): forward-model kernels against the oracle, map by map and gradient by gradient.
Prints a table instead of stopping at the first mismatch.  N=<images> (default 2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from ndivplanning_amd import _capi
from ndivplanning_amd.forward_trainer import ForwardModelTrainer
from ndivplanning_amd.models import forward_encoder as FE
from oracle import forward_model_oracle as FO

torch.set_num_threads(8)
n = int(os.environ.get("N", 2))
dev = "cuda:0"
state = FO.init_forward_model_state(int(os.environ.get("SEED", 5)))
model = FE.ForwardAutoencoder()
model.load_state_dict({k: v for k, v in state.items()}, strict=False)
gen = torch.Generator().manual_seed(int(os.environ.get("DATA_SEED", 6)))
frames = torch.rand(n, 3, 3, 128, 128, generator=gen) * 2.0 - 1.0
actions = torch.rand(n, 3, 4, generator=gen) * 2.0 - 1.0
cur, fut, act = frames[:, 0].contiguous(), frames[:, 1].contiguous(), actions[:, 0].contiguous()

# ---- oracle side: intermediates in fp64 through torch autograd on the mirror module's operator path
ref = FE.ForwardAutoencoder().double()
ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in model.state_dict().items()})
ref.train()
inter = {}


def hook(name):
    def fn(mod, inp, out):
        out.retain_grad()
        inter[name] = out
    return fn


for name in FE.LAYER_NAMES + FE.BN_NAMES:
    FE._module_tensor(ref, name).register_forward_hook(hook(name))
resid = ref(cur.double(), act.double())
loss = F.mse_loss(resid, (fut - cur).double())
loss.backward()
print("oracle loss %.8f" % loss.item())

# ---- HIP side
model = model.to(dev).train()
tr = ForwardModelTrainer(model, batch=n, keep_residual=True)
tr.grads(cur.to(dev), fut.to(dev), act.to(dev))
torch.cuda.synchronize()
print("hip    loss %.8f" % tr.loss.item())
lib = _capi.load()
names = ["COLS", "CAT6", "CAT5", "CAT4", "CAT3", "CAT2", "Z", "RAW1", "RAW2", "RAW3", "RAWU1", "RAWU2", "RAWU3", "RAWU4",
         "RAWU5", "RAWU6", "UP6", "RAWR1", "R1", "Y2", "DCAT6", "DCAT5", "DCAT4", "DCAT3", "DCAT2", "DZ", "DUP6", "DR1"]
shape = {"COLS": (64, 32), "CAT6": (64, 128), "CAT5": (32, 256), "CAT4": (16, 512), "CAT3": (8, 1024), "CAT2": (4, 2048),
         "RAW1": (64, 64), "RAW2": (32, 128), "RAW3": (16, 256), "RAWU1": (4, 1024), "RAWU2": (8, 512), "RAWU3": (16, 256),
         "RAWU4": (32, 128), "RAWU5": (64, 64), "RAWU6": (128, 32), "UP6": (128, 32), "RAWR1": (128, 32), "R1": (128, 32),
         "Y2": (128, 4), "DCAT6": (64, 128), "DCAT5": (32, 256), "DCAT4": (16, 512), "DCAT3": (8, 1024), "DCAT2": (4, 2048),
         "DUP6": (128, 32), "DR1": (128, 32)}


def ws(name):
    i = names.index(name)
    off = lib.ndp_fm_workspace_offset(n, i)
    if name in ("Z", "DZ"):
        return tr.workspace[off:off + n * 160].view(n, 160).cpu().double()
    h, c = shape[name]
    return tr.workspace[off:off + n * h * h * c].view(n, h, h, c).permute(0, 3, 1, 2).cpu().double()   # -> NCHW


def report(what, mine, want):
    err = (mine - want).abs().max().item()
    scale = want.abs().max().item()
    print("%-34s max|err| %.3e  scale %.3e  rel %.2e %s" % (what, err, scale, err / max(scale, 1e-30),
                                                           "" if err <= 2e-3 * scale + 1e-7 else "  <-- MISMATCH"))


print("---- forward maps (the backward overwrote raw maps with d raw; those are checked below)")
feats = [inter["encoder.conv%d_bn" % i].relu() for i in (1, 2, 3)] + [inter["encoder.conv4"].relu(), inter["encoder.conv5"].relu()]
ups = [inter["decoder.deconv%d_bn" % i].relu() for i in range(1, 7)]
for cat, up, feat in (("CAT6", ups[4], feats[0]), ("CAT5", ups[3], feats[1]), ("CAT4", ups[2], feats[2]),
                      ("CAT3", ups[1], feats[3]), ("CAT2", ups[0], feats[4])):
    c = up.shape[1]
    got = ws(cat)
    report(cat + "[skip half] = feat", got[:, c:], feat.detach())
    report(cat + "[up half]", got[:, :c], up.detach())
report("Z[:, :128] = code", ws("Z")[:, :128], inter["encoder.conv6"].detach().reshape(n, 128))
report("Z[:, 128:132] = action", ws("Z")[:, 128:132], act.double())
report("UP6", ws("UP6"), ups[5].detach())
report("R1[:, :16]", ws("R1")[:, :16], inter["decoder.conv_refine_1_bn"].relu().detach())
report("residual", tr.resid.cpu().double(), resid.detach())
print("---- d raw (pre-BatchNorm / pre-activation gradients), written over the raw maps")
for wsn, mod, c in (("RAWR1", "decoder.conv_refine_1", 16), ("RAWU6", "decoder.deconv6", 32), ("RAWU5", "decoder.deconv5", 64),
                    ("RAWU4", "decoder.deconv4", 128), ("RAWU3", "decoder.deconv3", 256), ("RAWU2", "decoder.deconv2", 512),
                    ("RAWU1", "decoder.deconv1", 1024), ("RAW3", "encoder.conv3", 256), ("RAW2", "encoder.conv2", 128),
                    ("RAW1", "encoder.conv1", 64)):
    report("d " + mod, ws(wsn)[:, :c], inter[mod].grad)
report("d code", ws("DZ")[:, :128], inter["encoder.conv6"].grad.reshape(n, 128))
print("---- d (post-ReLU map) * mask, from the d-cat buffers, against the gradient at the BatchNorm output")
for cat, dcat, lo, hi, mod in (("CAT4", "DCAT4", 256, 512, "encoder.conv3_bn"), ("CAT5", "DCAT5", 128, 256, "encoder.conv2_bn"),
                               ("CAT6", "DCAT6", 64, 128, "encoder.conv1_bn"), ("CAT4", "DCAT4", 0, 256, "decoder.deconv3_bn"),
                               ("CAT3", "DCAT3", 0, 512, "decoder.deconv2_bn")):
    y, dy = ws(cat)[:, lo:hi], ws(dcat)[:, lo:hi]
    report("%s[%d:%d] * mask" % (dcat, lo, hi), dy * (y > 0), inter[mod].grad)
print("---- gradients")
mine = tr.named_gradients()
for name, p in ref.named_parameters():
    if p.grad is None:
        continue
    report(name, mine[name].cpu().double(), p.grad)

# where the error of one tensor sits (FOCUS=<parameter name>): per slice along each of the first two dimensions
focus = os.environ.get("FOCUS")
if focus:
    g = dict(ref.named_parameters())[focus].grad
    e = (mine[focus].cpu().double() - g)
    print("focus", focus, "rel L2", (e.norm() / g.norm()).item())
    for dim in (0, 1):
        other = [d for d in range(e.dim()) if d != dim]
        en = (e * e).sum(dim=other).sqrt()
        gn = (g * g).sum(dim=other).sqrt()
        top = torch.topk(en, 5)
        print("  dim %d: top error slices" % dim, [(int(i), "%.2e" % v, "rel %.2e" % (v / gn[i])) for v, i in zip(top.values, top.indices)],
              "median slice error %.2e" % en.median().item())
