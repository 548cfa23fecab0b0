"""The oracle (oracle/gan_oracle.py) against the golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_params, load_golden
from oracle import gan_oracle as O

# free-running NDiv bound for steps >= 1 at FLAT >= 21 (measured 1e-3); the B=2 fixture
# is only compared per step
FREE_RUN_NDIV_RTOL = 2e-2
FREE_RUN_CASES = ["step_cfg1", "step_dsteps2_nz5", "step_k32"]
STEP_CASES = ["step_tiny_full", "step_cfg1", "step_dsteps2_nz5", "step_k32"]


def _meta(rec):
    seed, batch, k, nz, steps, dsteps, traj = [int(v) for v in rec["meta"]]
    return dict(seed=seed, batch=batch, k=k, nz=nz, steps=steps, dsteps=dsteps, traj=traj,
                factor=float(rec["factor"]), lr=float(rec["lr"]))


def _close(a, b, atol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= atol, "%s: max |diff| %.3e > %.1e" % (what, err, atol)


def _params_close(p, p_ref, what, lr, steps, g_ref=None, g_scale=1.0):
    """Post-Adam parameters.  Adam's first steps move a weight by ~lr*sign(g), so an
    element whose gradient is below the fp32 noise floor of the gradient itself
    (cancellation among terms of the net's gradient scale; summation order decides the
    sign) legitimately lands up to 2*lr per step away.  Gate: every element within the
    Adam bound; beyond 1e-4 only (where the reference gradient is known) at gradients
    under the noise floor 2e-6 * max(1, net-wide max |g|), else at most 6 % of a tensor."""
    p = np.asarray(p, dtype=np.float64)
    p_ref = np.asarray(p_ref, dtype=np.float64)
    err = np.abs(p - p_ref)
    assert err.max() <= 2.5 * lr * steps, "%s: max |diff| %.3e beyond the Adam bound" % (what, err.max())
    bad = err > 1e-4
    if not bad.any():
        return
    if g_ref is not None:
        floor = 2e-6 * max(1.0, g_scale)
        assert np.abs(np.asarray(g_ref)[bad]).max() <= floor, \
            "%s: mismatch at a well-conditioned gradient (|g| %.2e > floor %.2e)" % (
                what, np.abs(np.asarray(g_ref)[bad]).max(), floor)
    else:
        assert bad.mean() <= 0.06 or bad.sum() <= 1, "%s: %.2f %% of elements differ by > 1e-4" % (what, 100 * bad.mean())


def _gscale(grads):
    return max(float(np.abs(np.asarray(g)).max()) for g in grads.values())


def _ndiv_close(a, ref, what):
    # the NDiv sum is compared with |d| <= 1e-4 * max(1, |ref|)
    assert abs(float(a) - float(ref)) <= 1e-4 * max(1.0, abs(float(ref))), what


@pytest.mark.parametrize("case", STEP_CASES)
def test_init_params_match_reference(case):
    rec = load_golden(case)
    m = _meta(rec)
    g, d = O.init_params(m["seed"], m["nz"])
    for n, p in golden_params(rec, "g0.").items():
        assert torch.equal(g[n], p), n
    for n, p in golden_params(rec, "d0.").items():
        assert torch.equal(d[n], p), n


@pytest.mark.parametrize("case", FREE_RUN_CASES)
def test_autograd_restatement_matches_reference_free_running(case):
    """The restated loop (same torch ops, same order) against the reference through
    every step without forcing.  Step 0 is tight.  Later steps depend on Adam's
    amplification of summation-order noise (see test_manual_step_teacher_forced); even
    torch against itself at another thread count moves NDiv by 1e-3 there, so they are
    gated at the looser free-running bounds."""
    rec = load_golden(case)
    m = _meta(rec)
    g, d = golden_params(rec, "g0."), golden_params(rec, "d0.")
    codes, actions = torch.from_numpy(rec["codes"]), torch.from_numpy(rec["actions"])
    noise = torch.from_numpy(rec["noise"])
    tr = O.AutogradTrainer(g, d, lr=m["lr"], pairwise_div_factor=m["factor"])
    for s in range(m["steps"]):
        out = tr.step(codes, actions, noise[s], discrim_steps=m["dsteps"])
        d_loss, g_loss, pd = rec["s%d.losses" % s]
        tol = 1e-6 if s == 0 else 1e-5
        _close(out["d_loss"], d_loss, tol, "D_loss step %d" % s)
        _close(out["g_loss"], g_loss, tol, "G_loss step %d" % s)
        if s == 0:
            _ndiv_close(out["pair_div"], pd, "pair_div step 0")
        else:
            assert abs(float(out["pair_div"]) - pd) <= FREE_RUN_NDIV_RTOL * abs(pd), "pair_div step %d" % s
        if s == 0:
            for key in ("action_hat", "logits_real", "logits_fake", "logits_gen"):
                _close(out[key], rec["s0." + key], 1e-6, key)
            for kind, grads in (("dgrad", out["d_grads"]), ("ggrad", out["g_grads"])):
                for n, gr in grads.items():
                    key = "s0.%s.%s" % (kind, n)
                    if key in rec:
                        _close(gr, rec[key], 5e-6, key)
        gp, dp = tr.params()
        if "s%d.g.fc1.weight" % s in rec:
            for n, p in golden_params(rec, "s%d.g." % s).items():
                _params_close(gp[n], p, "G %s after step %d" % (n, s), m["lr"], s + 1,
                              rec.get("s0.ggrad." + n) if s == 0 else None, 30.0)
            for n, p in golden_params(rec, "s%d.d." % s).items():
                _params_close(dp[n], p, "D %s after step %d" % (n, s), m["lr"], (s + 1) * m["dsteps"],
                              rec.get("s0.dgrad." + n) if s == 0 and m["dsteps"] == 1 else None)


@pytest.mark.parametrize("case", STEP_CASES)
def test_manual_step_teacher_forced(case):
    """The hand-written step (the formulation the HIP kernels implement) against the
    reference, one step at a time from the reference's exact state.  Forcing is
    needed because Adam amplifies gradient noise: an element whose gradient is below
    the fp32 noise floor moves by +-lr according to summation order, and the NDiv
    loss of the NEXT step is sensitive to that (measured: 1e-3 relative at B=16,
    15 % at B=2), so no independent fp32 implementation can track NDiv free-running
    to 1e-4.  Per step, from identical state, everything matches tightly."""
    rec = load_golden(case)
    m = _meta(rec)
    g, d = golden_params(rec, "g0."), golden_params(rec, "d0.")
    codes, actions = torch.from_numpy(rec["codes"]), torch.from_numpy(rec["actions"])
    noise = torch.from_numpy(rec["noise"])
    teacher = O.AutogradTrainer(g, d, lr=m["lr"], pairwise_div_factor=m["factor"])
    sm = O.StepMath(golden_params(rec, "g0."), golden_params(rec, "d0."), lr=m["lr"],
                    pairwise_div_factor=m["factor"])
    for s in range(m["steps"]):
        sm.load_state(teacher.export_state())
        ref = teacher.step(codes, actions, noise[s], discrim_steps=m["dsteps"])
        out = sm.step(codes, actions, noise[s], discrim_steps=m["dsteps"])
        d_loss, g_loss, pd = rec["s%d.losses" % s]
        _close(out["d_loss"], d_loss, 1e-5, "D_loss step %d" % s)
        _close(out["g_loss"], g_loss, 1e-5, "G_loss step %d" % s)
        _ndiv_close(out["pair_div"], pd, "pair_div step %d" % s)
        for key in ("action_hat", "logits_real", "logits_fake", "logits_gen"):
            _close(out[key], ref[key], 1e-5, "%s step %d" % (key, s))
        gp_ref, dp_ref = teacher.params()
        for n in sm.g:
            _params_close(sm.g[n], gp_ref[n], "G %s after step %d" % (n, s), m["lr"], 1,
                          ref["g_grads"][n].numpy(), _gscale(ref["g_grads"]))
        for n in sm.d:
            _params_close(sm.d[n], dp_ref[n], "D %s after step %d" % (n, s), m["lr"], m["dsteps"],
                          ref["d_grads"][n].numpy() if m["dsteps"] == 1 else None, _gscale(ref["d_grads"]))


def test_manual_gradients_match_reference_gradients():
    """The hand-written backward (what the HIP kernels implement) against the
    reference's autograd gradients, step 0 of the full-detail fixture."""
    rec = load_golden("step_tiny_full")
    m = _meta(rec)
    g, d = golden_params(rec, "g0."), golden_params(rec, "d0.")
    sm = O.StepMath(g, d, lr=m["lr"], pairwise_div_factor=m["factor"])
    codes, actions = torch.from_numpy(rec["codes"]), torch.from_numpy(rec["actions"])
    noise = torch.from_numpy(rec["noise"])[0]
    sm.g_forward(codes, actions, noise)
    dg = sm.d_grads()
    # gradients: |d| <= 1e-5 * max(1, net-wide max |g|)  (fp32 cancellation noise scales
    # with the summands; the G gradients carry NDiv terms of O(1..30))
    dscale = max(1.0, max(np.abs(rec["s0.dgrad." + n]).max() for n in dg))
    for n, gr in dg.items():
        _close(gr, rec["s0.dgrad." + n], 1e-5 * dscale, "D grad " + n)
    sm.apply_d(dg)
    gg = sm.g_grads()
    gscale = max(1.0, max(np.abs(rec["s0.ggrad." + n]).max() for n in gg))
    for n, gr in gg.items():
        _close(gr, rec["s0.ggrad." + n], 1e-5 * gscale, "G grad " + n)


def test_cfg2_scalars():
    """BASELINE config 2 shape (B=64, K=6): losses of 3 steps, inputs regenerated
    from the seeds the generator used."""
    rec = load_golden("step_cfg2_scalars")
    m = _meta(rec)
    g, d = O.init_params(m["seed"], m["nz"])
    gen = torch.Generator().manual_seed(m["seed"] + 1000)
    flat = m["batch"] * (m["traj"] - 1)
    codes = torch.randn(flat, 256, generator=gen)
    actions = torch.rand(flat, 4, generator=gen) * 2.0 - 1.0
    noise = torch.rand(m["steps"], flat, m["k"], m["nz"], generator=gen)
    sm = O.StepMath(g, d, lr=m["lr"], pairwise_div_factor=m["factor"])
    for s in range(m["steps"]):
        out = sm.step(codes, actions, noise[s])
        if s == 0:
            ah = out["action_hat"].double()
            _close([ah.sum(), ah.abs().sum()], rec["s0.action_hat_sum"], 1e-2, "action_hat sums")
        d_loss, g_loss, pd = rec["s%d.losses" % s]
        _close(out["d_loss"], d_loss, 1e-5, "D_loss")
        _close(out["g_loss"], g_loss, 1e-5, "G_loss")
        if s == 0:
            _ndiv_close(out["pair_div"], pd, "pair_div")
        else:   # free-running, see test_manual_step_teacher_forced
            assert abs(float(out["pair_div"]) - pd) <= FREE_RUN_NDIV_RTOL * abs(pd)


@pytest.mark.parametrize("name", ["k6", "k32", "k2", "k3c5", "coin"])
def test_ndiv_loss_and_grad(name):
    rec = load_golden("ndiv_cases")
    x, z = torch.from_numpy(rec[name + ".x"]), torch.from_numpy(rec[name + ".z"])
    loss, grad = O.ndiv_loss_and_grad(x, z)
    _ndiv_close(loss, rec[name + ".loss"], name)
    _close(grad, rec[name + ".grad"], 1e-5, name + " grad")
    _ndiv_close(O.compute_pairwise_divergence(x, z), rec[name + ".loss"], name)
    if name + ".pair_x" in rec:
        _close(O.compute_pair_distance(x), rec[name + ".pair_x"], 1e-6, "pair distance")
        _close(O.compute_pairwise(x), rec[name + ".pairwise_x"], 1e-6, "pairwise")


def test_ndiv_k1_is_nan_like_reference():
    rec = load_golden("ndiv_cases")
    assert np.isnan(rec["k1.loss"])
    loss = O.compute_pairwise_divergence(torch.from_numpy(rec["k1.x"]), torch.from_numpy(rec["k1.z"]))
    assert torch.isnan(loss)


def test_ndiv_grad_fp64_finite_differences():
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, 4, generator=gen, dtype=torch.float64)
    z = torch.rand(2, 5, 2, generator=gen, dtype=torch.float64)
    _, grad = O.ndiv_loss_and_grad(x, z)
    # the analytic gradient treats the row sums as constants (diversity.py:18), so
    # difference the loss with the denominators frozen at x
    dx0 = O.compute_pairwise(x)
    sx = dx0.sum(dim=2, keepdim=True)
    zt = O.compute_pair_distance(z)

    def frozen(xx):
        return torch.clamp_min(0.8 * zt - O.compute_pairwise(xx) / sx, 0).sum()

    eps = 1e-6
    num = torch.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        xp, xm = x.clone(), x.clone()
        xp[idx] += eps
        xm[idx] -= eps
        num[idx] = (frozen(xp) - frozen(xm)) / (2 * eps)
    assert (num - grad).abs().max() < 1e-6
