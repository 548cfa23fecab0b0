"""CPU: the restatement of the reference's forward (next-frame) model and of its training iteration
(oracle/forward_model_oracle.py) against the golden vector generated from the reference's
own ForwardAutoencoder + MSELoss + Adam (tests/golden/make_golden_forward_model.py)."""
import numpy as np
import torch

from conftest import load_golden
from oracle import forward_model_oracle as FO


def _sums(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def _inputs(data_seed, n):
    gen = torch.Generator().manual_seed(data_seed)
    frames = torch.rand(n, 3, 3, 128, 128, generator=gen) * 2.0 - 1.0
    actions = torch.rand(n, 3, 4, generator=gen) * 2.0 - 1.0
    return frames, actions


import pytest


@pytest.mark.parametrize("case", ["forward_model_case", "forward_model_case_sharp"])
def test_two_training_iterations_match_the_reference(case):
    g = load_golden(case)
    seed, data_seed, n = (int(v) for v in g["meta"])
    torch.set_num_threads(1)
    state = FO.init_forward_model_state(seed)
    w64 = torch.cat([v.double().reshape(-1) for v in _ref_order(state).values() if v.dtype.is_floating_point])
    # the fixture stores seeds, not 33 M weights: the RNG stream must have produced the same initial state
    np.testing.assert_allclose(np.array([w64.sum().item(), w64.abs().sum().item(), float(w64.numel())]),
                               g["state_checksum"], rtol=1e-12)
    names = [str(s) for s in g["names"]]
    assert FO.trainable(state) == names                  # optimizer order: decoder group, then encoder group
    tr = FO.ForwardModelTrainer(state, lr=float(g["lr"]))
    frames, actions = _inputs(data_seed, n)
    for it in range(2):
        out = tr.step(frames[:, it], frames[:, it + 1], actions[:, it])
        assert abs(out["loss"].item() - float(g["s%d.loss" % it])) <= 1e-6
        np.testing.assert_allclose(out["resid"][:, :, ::16, ::16].numpy(), g["s%d.resid_sample" % it], atol=2e-6)
        np.testing.assert_allclose(_sums(out["resid"]), g["s%d.resid_sums" % it], rtol=1e-5)
        for i, name in enumerate(names):
            grad = out["grads"][name]
            want = g["s%d.grad_sums" % it][i]
            if grad is None:                              # conv4_bn / conv5_bn: constructed, never applied
                assert not want.any(), name
                continue
            scale = max(want[1], 1e-12)
            assert abs(_sums(grad)[1] - want[1]) <= 2e-4 * scale, (name, it)
            # Adam's first steps move every element by ~lr whatever the gradient's size: compare the parameters
            got = _sums(state[name])
            wantp = g["s%d.param_sums" % it][i]
            assert abs(got[1] - wantp[1]) <= 1e-5 * max(wantp[1], 1e-12) + 1e-7, (name, it)
    running = np.stack([_sums(state[k]) for k in _ref_order(state) if "running_" in k])
    np.testing.assert_allclose(running, g["running_sums"], rtol=1e-5, atol=1e-7)


def test_eval_mode_adds_the_residual_and_uses_running_statistics():
    state = FO.init_forward_model_state(3)
    frames, actions = _inputs(4, 1)
    with torch.no_grad():
        a = FO.forward(state, frames[:, 0], actions[:, 0], training=False)
        r = FO.forward({k: v.clone() for k, v in state.items()}, frames[:, 0], actions[:, 0], training=True)
    assert a.shape == (1, 3, 128, 128) and r.shape == (1, 3, 128, 128)
    assert float((a - frames[:, 0]).abs().max()) <= 1.0 + 1e-6       # |tanh| <= 1
    assert int(state["decoder.deconv1_bn.num_batches_tracked"]) == 0  # eval mode left the statistics alone


def _ref_order(state):
    """state_dict order of the reference module: encoder.*, then decoder.* (forward_encoder.py:101-103)."""
    return state
