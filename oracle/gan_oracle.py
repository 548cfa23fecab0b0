"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- CPU restatement, in plain
PyTorch-CPU tensor arithmetic, of the reference's GAN train step.

Every function cites the reference lines it restates (paths relative to the
reference checkout).  Two independent formulations are kept so that they check
each other and so that HIP kernels can be compared stage by stage:

* ``train_step_autograd``  -- the reference's step order driven by torch
  autograd + torch.optim.Adam (train_gan.py:159-203), wasted work included.
* ``StepMath``             -- the same step with every gradient written out
  by hand (autograd-minimal), split into the four phases the HIP path and the
  data-parallel driver use: d_grads / apply_d / g_grads / apply_g.

dtype is a parameter so the same code runs in fp64 to adjudicate summation
order differences.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

CODE_DIM = 256          # cat(state_code, target_code): train_gan.py:155
ACTION_DIM = 4          # models/gan.py:71, 94
LRELU_SLOPE = 0.01      # F.leaky_relu default, models/gan.py:106-108
HINGE_ALPHA = 0.8       # diversity.py:40

ParamDict = Dict[str, torch.Tensor]


def g_layer_dims(noise_dim: int) -> List[Tuple[int, int]]:
    """(in, out) of Decoder.fc1..fc5, models/gan.py:67-71."""
    return [(CODE_DIM + noise_dim, 128), (128, 64), (64, 128), (128, 256), (256, ACTION_DIM)]


def d_layer_dims() -> List[Tuple[int, int]]:
    """(in, out) of Discriminator.fc1..fc4, models/gan.py:94-97."""
    return [(CODE_DIM + ACTION_DIM, 64), (64, 128), (128, 256), (256, 1)]


def init_params(seed: int, noise_dim: int, dtype=torch.float32) -> Tuple[ParamDict, ParamDict]:
    """Initial G and D parameters exactly as the reference obtains them:
    torch.manual_seed (train_gan.py:65), then Decoder(...) then Discriminator()
    constructed in that order (train_gan.py:89-90); weight_init is a no-op for
    nn.Linear (models/gan.py:15-18, 74-76), so torch's default Linear init stays.
    Keys are the reference state_dict keys (fc1.weight ...)."""
    torch.manual_seed(seed)
    nets = []
    for dims in (g_layer_dims(noise_dim), d_layer_dims()):
        p: ParamDict = OrderedDict()
        for i, (fin, fout) in enumerate(dims, start=1):
            lin = torch.nn.Linear(fin, fout)
            p["fc%d.weight" % i] = lin.weight.detach().to(dtype).clone()
            p["fc%d.bias" % i] = lin.bias.detach().to(dtype).clone()
        nets.append(p)
    return nets[0], nets[1]


def _n_layers(p: ParamDict) -> int:
    return len(p) // 2


# --------------------------------------------------------------------------- forward

def g_forward(g: ParamDict, z: torch.Tensor, keep: bool = False):
    """Decoder.forward, models/gan.py:79-86: relu after fc1..fc4, none after fc5.
    Returns action_hat, or (action_hat, [z, h1, h2, h3, h4]) with keep=True."""
    acts = [z]
    h = z
    n = _n_layers(g)
    for i in range(1, n + 1):
        h = h @ g["fc%d.weight" % i].t() + g["fc%d.bias" % i]
        if i < n:
            h = torch.clamp_min(h, 0.0)
            acts.append(h)
    return (h, acts) if keep else h


def d_forward(d: ParamDict, action: torch.Tensor, code: torch.Tensor, keep: bool = False):
    """Discriminator.forward, models/gan.py:104-110: cat([action, code]) then
    leaky_relu(0.01) after fc1..fc3, none after fc4.  Returns logits [M,1]."""
    x = torch.cat([action, code], dim=1)
    acts = [x]
    h = x
    n = _n_layers(d)
    for i in range(1, n + 1):
        h = h @ d["fc%d.weight" % i].t() + d["fc%d.bias" % i]
        if i < n:
            h = torch.where(h > 0, h, h * LRELU_SLOPE)
            acts.append(h)
    return (h, acts) if keep else h


def bce_with_logits_sum(x: torch.Tensor, target: float) -> torch.Tensor:
    """Sum over elements of nn.BCEWithLogitsLoss's per-element term
    max(x,0) - x*y + log1p(exp(-|x|)) (train_gan.py:174-190 use the mean)."""
    return (torch.clamp_min(x, 0.0) - x * target + torch.log1p(torch.exp(-x.abs()))).sum()


def bce_with_logits_grad(x: torch.Tensor, target: float, inv_count: float) -> torch.Tensor:
    """d(mean BCE)/dx = (sigmoid(x) - y) / count."""
    return (torch.sigmoid(x) - target) * inv_count


# --------------------------------------------------------------------------- NDiv

def compute_pairwise(z: torch.Tensor) -> torch.Tensor:
    """diversity.py:8-9: all-pairs L2 distance inside each row n; [N,K,C]->[N,K,K]."""
    diff = z[:, :, None, :] - z[:, None, :, :]
    # vector_norm (what torch.norm(p=2) dispatches to) so that autograd masks the
    # 0/0 sub-gradient at coincident samples and on the diagonal, as in the reference
    return torch.linalg.vector_norm(diff, ord=2, dim=3)


def compute_pair_distance(z: torch.Tensor) -> torch.Tensor:
    """diversity.py:12-19 with weight=None: divide by the row sum (detached)."""
    d = compute_pairwise(z)
    return d / d.sum(dim=2, keepdim=True).detach()


def compute_pairwise_divergence(recodes: torch.Tensor, codes: torch.Tensor) -> torch.Tensor:
    """diversity.py:36-41: sum relu(0.8*z_tilde - x_tilde)."""
    n, k = codes.shape[0], codes.shape[1]
    zt = compute_pair_distance(codes.reshape(n, k, -1))
    xt = compute_pair_distance(recodes.reshape(n, k, -1))
    return torch.clamp_min(zt * HINGE_ALPHA - xt, 0.0).sum()


def ndiv_loss_and_grad(x: torch.Tensor, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Loss of compute_pairwise_divergence(x, z) and its gradient w.r.t. x written
    out by hand:
      dL/dx_i = sum_j -(m_ij/s_i + m_ji/s_j) (x_i - x_j)/d_ij,  0 where d_ij == 0
    m_ij = 1[0.8 z~_ij - x~_ij > 0]; s_i = sum_j d_ij is treated as a constant
    (diversity.py:18 detaches it); torch.norm's backward is 0 at d == 0."""
    dx = compute_pairwise(x)
    dz = compute_pairwise(z)
    sx = dx.sum(dim=2, keepdim=True)
    sz = dz.sum(dim=2, keepdim=True)
    h = HINGE_ALPHA * dz / sz - dx / sx
    loss = torch.clamp_min(h, 0.0).sum()
    m = (h > 0).to(x.dtype)
    w = m / sx                                   # m_ij / s_i
    w = w + w.transpose(1, 2)                    # + m_ji / s_j
    inv_d = torch.where(dx > 0, 1.0 / dx, torch.zeros_like(dx))
    diff = x[:, :, None, :] - x[:, None, :, :]
    grad = -((w * inv_d)[..., None] * diff).sum(dim=2)
    return loss, grad


# --------------------------------------------------------------------------- Adam

class AdamState:
    """torch.optim.Adam state for one network (train_gan.py:98-104): lr 2e-4,
    betas (0.5, 0.999), eps 1e-8, no weight decay, no amsgrad."""

    def __init__(self, params: ParamDict, lr: float, betas=(0.5, 0.999), eps: float = 1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = 0
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())

    def apply(self, params: ParamDict, grads: ParamDict) -> None:
        b1, b2 = self.betas
        self.t += 1
        bc1 = 1.0 - b1 ** self.t
        bc2 = 1.0 - b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for k, p in params.items():
            g = grads[k]
            m, v = self.m[k], self.v[k]
            m += (g - m) * (1.0 - b1)                      # exp_avg.lerp_(grad, 1-beta1)
            v.mul_(b2).add_(g * g * (1.0 - b2))            # exp_avg_sq.mul_().addcmul_()
            denom = v.sqrt() / bc2_sqrt + self.eps
            p -= step_size * (m / denom)


# --------------------------------------------------------------------------- step, by hand

def make_generator_input(codes: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """diverse_sampling's concat, train_gan.py:42-47 and 165: [FLAT,K,256+nz] -> [M,256+nz]."""
    flat, k = noise.shape[0], noise.shape[1]
    z = torch.cat([codes[:, None, :].expand(-1, k, -1), noise], dim=2)
    return z.reshape(flat * k, -1)


def _mlp_backward(p: ParamDict, acts: List[torch.Tensor], dy_last: torch.Tensor, kind: str,
                  need_dx: bool) -> Tuple[ParamDict, Optional[torch.Tensor]]:
    """Backward of an MLP given saved layer inputs ``acts`` (acts[l] is the input of
    fc(l+1), post-activation) and the gradient at the last layer's output.
    kind: 'relu' (G) or 'lrelu' (D).  Returns (param grads, grad at network input)."""
    n = _n_layers(p)
    grads: ParamDict = OrderedDict()
    dy = dy_last
    dx = None
    for i in range(n, 0, -1):
        x = acts[i - 1]
        grads["fc%d.weight" % i] = dy.t() @ x
        grads["fc%d.bias" % i] = dy.sum(dim=0)
        if i > 1 or need_dx:
            dx = dy @ p["fc%d.weight" % i]
        if i > 1:
            if kind == "relu":
                dy = dx * (x > 0).to(dx.dtype)           # threshold_backward: 0 at x == 0
            else:
                dy = torch.where(x > 0, dx, dx * LRELU_SLOPE)
    ordered = OrderedDict((k, grads[k]) for k in p.keys())
    return ordered, dx


class StepMath:
    """One GAN train step (train_gan.py:159-203) in four phases, gradients by hand.

    ``inv_m`` is 1/M_global so that a rank holding a shard of the batch produces
    its share of the *global* BCE mean; NDiv is a sum and needs no scaling
    (SURVEY.md section 8e).  With inv_m = 1/M_local this is the single-process step."""

    def __init__(self, g: ParamDict, d: ParamDict, lr: float = 2e-4,
                 pairwise_div_factor: float = 0.1):
        self.g, self.d = g, d
        self.g_opt = AdamState(g, lr)
        self.d_opt = AdamState(d, lr)
        self.factor = pairwise_div_factor
        self.out: Dict[str, torch.Tensor] = {}

    def load_state(self, state: Dict[str, object]) -> None:
        """Teacher-force to an AutogradTrainer.export_state() snapshot."""
        for dst, src in ((self.g, state["g"]), (self.d, state["d"])):
            for k in dst:
                dst[k].copy_(src[k])
        for opt, src in ((self.g_opt, state["g_opt"]), (self.d_opt, state["d_opt"])):
            opt.t = src["t"]
            for k in opt.m:
                opt.m[k].copy_(src["m"][k])
                opt.v[k].copy_(src["v"][k])

    # phase 0 + 1: G forward, D loss and D gradients
    def g_forward(self, codes, actions, noise):
        flat, k = noise.shape[0], noise.shape[1]
        self.codes, self.noise = codes, noise
        self.codes_rep = torch.repeat_interleave(codes, k, dim=0)       # train_gan.py:156
        self.actions_rep = torch.repeat_interleave(actions, k, dim=0)   # train_gan.py:140
        z = make_generator_input(codes, noise)
        self.action_hat, self.g_acts = g_forward(self.g, z, keep=True)
        self.out["action_hat"] = self.action_hat
        return self.action_hat

    def d_grads(self, inv_m: Optional[float] = None) -> ParamDict:
        m_rows = self.action_hat.shape[0]
        inv_m = 1.0 / m_rows if inv_m is None else inv_m
        lr_, acts_r = d_forward(self.d, self.actions_rep, self.codes_rep, keep=True)
        lf_, acts_f = d_forward(self.d, self.action_hat, self.codes_rep, keep=True)
        d_loss = (bce_with_logits_sum(lr_, 1.0) + bce_with_logits_sum(lf_, 0.0)) * inv_m
        gr, _ = _mlp_backward(self.d, acts_r, bce_with_logits_grad(lr_, 1.0, inv_m), "lrelu", False)
        gf, _ = _mlp_backward(self.d, acts_f, bce_with_logits_grad(lf_, 0.0, inv_m), "lrelu", False)
        self.out.update(logits_real=lr_, logits_fake=lf_, d_loss=d_loss)
        return OrderedDict((k_, gr[k_] + gf[k_]) for k_ in gr)

    def apply_d(self, grads: ParamDict) -> None:
        self.d_opt.apply(self.d, grads)

    # phase 2: G loss (through the *updated* D, train_gan.py:184 -> 187) + NDiv
    def g_grads(self, inv_m: Optional[float] = None) -> ParamDict:
        m_rows = self.action_hat.shape[0]
        inv_m = 1.0 / m_rows if inv_m is None else inv_m
        flat, k = self.noise.shape[0], self.noise.shape[1]
        lg, acts = d_forward(self.d, self.action_hat, self.codes_rep, keep=True)
        g_loss = bce_with_logits_sum(lg, 1.0) * inv_m
        _, dx = _mlp_backward(self.d, acts, bce_with_logits_grad(lg, 1.0, inv_m), "lrelu", True)
        d_action = dx[:, :ACTION_DIM]
        pd, nd_grad = ndiv_loss_and_grad(self.action_hat.reshape(flat, k, -1), self.noise)
        d_action = d_action + self.factor * nd_grad.reshape(m_rows, -1)
        grads, _ = _mlp_backward(self.g, self.g_acts, d_action, "relu", False)
        self.out.update(logits_gen=lg, g_loss=g_loss, pair_div=pd, d_action=d_action)
        return grads

    def apply_g(self, grads: ParamDict) -> None:
        self.g_opt.apply(self.g, grads)

    def step(self, codes, actions, noise, discrim_steps: int = 1) -> Dict[str, torch.Tensor]:
        self.g_forward(codes, actions, noise)
        for _ in range(discrim_steps):                       # train_gan.py:172-184
            self.apply_d(self.d_grads())
        self.apply_g(self.g_grads())
        return dict(self.out)


# --------------------------------------------------------------------------- step, autograd

class AutogradTrainer:
    """The reference's loop body (train_gan.py:159-203) on plain tensors with
    torch autograd + torch.optim.Adam: same op order, same retained graph, same
    zero_grad placement.  This is also the CPU baseline that bench.py times."""

    def __init__(self, g: ParamDict, d: ParamDict, lr: float = 2e-4,
                 pairwise_div_factor: float = 0.1):
        self.g = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in g.items())
        self.d = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in d.items())
        self.g_opt = torch.optim.Adam(list(self.g.values()), lr=lr, betas=(0.5, 0.999))
        self.d_opt = torch.optim.Adam(list(self.d.values()), lr=lr, betas=(0.5, 0.999))
        self.factor = pairwise_div_factor
        self.bce = torch.nn.BCEWithLogitsLoss()

    def step(self, codes, actions, noise, discrim_steps: int = 1) -> Dict[str, torch.Tensor]:
        flat, k = noise.shape[0], noise.shape[1]
        m_rows = flat * k
        action_rep = torch.repeat_interleave(actions, k, dim=0)
        codes_rep = torch.repeat_interleave(codes, k, dim=0)
        z = make_generator_input(codes, noise)
        action_hat = g_forward(self.g, z)
        ones, zeros = torch.ones(m_rows, dtype=z.dtype), torch.zeros(m_rows, dtype=z.dtype)
        out = {"action_hat": action_hat.detach().clone()}
        for _ in range(discrim_steps):
            lr_ = d_forward(self.d, action_rep, codes_rep)
            lf_ = d_forward(self.d, action_hat, codes_rep)
            d_loss = self.bce(lr_.squeeze(1), ones) + self.bce(lf_.squeeze(1), zeros)
            self.d_opt.zero_grad()
            d_loss.backward(retain_graph=True)
            out.update(d_loss=d_loss.detach().clone(), logits_real=lr_.detach().clone(),
                       logits_fake=lf_.detach().clone(),
                       d_grads=OrderedDict((n, p.grad.clone()) for n, p in self.d.items()))
            self.d_opt.step()
        lg = d_forward(self.d, action_hat, codes_rep)
        g_loss = self.bce(lg.squeeze(1), ones)
        pair_div = compute_pairwise_divergence(action_hat.view(flat, k, -1), noise)
        total = g_loss + self.factor * pair_div
        self.g_opt.zero_grad()
        total.backward()
        out.update(g_loss=g_loss.detach().clone(), pair_div=pair_div.detach().clone(),
                   logits_gen=lg.detach().clone(),
                   g_grads=OrderedDict((n, p.grad.clone()) for n, p in self.g.items()))
        self.g_opt.step()
        return out

    def params(self) -> Tuple[ParamDict, ParamDict]:
        return (OrderedDict((k, v.detach().clone()) for k, v in self.g.items()),
                OrderedDict((k, v.detach().clone()) for k, v in self.d.items()))

    def export_state(self) -> Dict[str, object]:
        """Parameters and Adam moments/step counts, for teacher-forcing another
        implementation to the exact state the reference arithmetic is in."""
        def opt_state(opt, params):
            m, v, t = OrderedDict(), OrderedDict(), 0
            for name, p in params.items():
                st = opt.state.get(p, None)
                if st:
                    m[name], v[name] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
                    t = int(st["step"])
                else:
                    m[name], v[name] = torch.zeros_like(p), torch.zeros_like(p)
            return {"m": m, "v": v, "t": t}
        g, d = self.params()
        return {"g": g, "d": d, "g_opt": opt_state(self.g_opt, self.g), "d_opt": opt_state(self.d_opt, self.d)}


# --------------------------------------------------------------------------- synthetic batches

def synthetic_batch(seed: int, batch: int, num_sample: int, noise_dim: int = 2,
                    traj_len: int = 8, steps: int = 1, dtype=torch.float32):
    """Codes-mode synthetic inputs of SURVEY.md section 8d / BASELINE.md section 2:
    codes ~ N(0,1) [FLAT,256], actions ~ U[-1,1) [FLAT,4], noise ~ U[0,1) [steps,FLAT,K,nz],
    all from a seeded CPU generator."""
    gen = torch.Generator().manual_seed(seed)
    flat = batch * (traj_len - 1)
    codes = torch.randn(flat, CODE_DIM, generator=gen).to(dtype)
    actions = (torch.rand(flat, ACTION_DIM, generator=gen) * 2.0 - 1.0).to(dtype)
    noise = torch.rand(steps, flat, num_sample, noise_dim, generator=gen).to(dtype)
    return codes, actions, noise
