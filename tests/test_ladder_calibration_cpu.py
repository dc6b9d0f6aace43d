"""CPU: the arithmetic of the precision ladder's self-calibration (hip_ext.engine.ladder_curve / ladder_thresholds: pure torch) on synthetic logits whose error is
known -- what DepthEngine.calibrate computes on the device from the rungs' logits.  No GPU, no library call."""
import math

import pytest
import torch

from hip_ext import ACT_NONE, ACT_RELU, ACT_SIGMOID
from hip_ext.engine import ladder_curve, ladder_thresholds


def _logits(n=4, p=40000, seed=0, spread=2.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 1, 200, p // 200, generator=g) * spread


def test_bare_logits_the_threshold_is_budget_over_the_mean_logit_error():
    z3 = _logits(seed=1)
    noise = torch.randn(z3.shape, generator=torch.Generator().manual_seed(2)) * 1e-3
    R, E = ladder_curve(z3 + noise, z3, ACT_NONE)
    assert R.shape == (4, 1) and torch.allclose(E / R, noise.abs().flatten(1).double().mean(1, keepdim=True), rtol=1e-4)
    res = ladder_thresholds(z3 + noise, None, z3, ACT_NONE, budget=9e-4, safety=1.1, rule="global")
    eps = float(noise.abs().flatten(1).double().mean(1).max())
    assert res["eps1"] == pytest.approx(eps, rel=1e-4) and res["r3"] == pytest.approx(9e-4 / (1.1 * eps), rel=1e-4) and "r" not in res   # (z3 + noise is formed in fp32)


@pytest.mark.parametrize("act", [ACT_SIGMOID, ACT_RELU])
def test_eps_is_the_sensitivity_weighted_logit_error_and_the_rules_order(act):
    """A logit error of constant magnitude delta with random sign: e / r = delta at every operating point (to first order), so global = budget / (safety delta);
    the cross rule leaves the rung at the first grid point whose metric exceeds budget / safety -- never below the global threshold."""
    z3 = _logits(seed=3)
    delta = 2e-3
    sign = torch.randint(0, 2, z3.shape, generator=torch.Generator().manual_seed(4)).float() * 2 - 1
    z1 = z3 + delta * sign
    R, E = ladder_curve(z1, z3, act)
    assert R.shape[0] == 4 and R.shape[1] >= 11 and bool((R > 0).all())
    ratio = (E / R)
    assert float(ratio.max()) == pytest.approx(delta, rel=0.03) and float(ratio.min()) == pytest.approx(delta, rel=0.05)
    if act == ACT_SIGMOID:
        assert float(R.min()) < 0.2 and float(R.max()) > 0.9           # the grid spans centred maps to maps near 0
    g = ladder_thresholds(z1, None, z3, act, budget=9e-4, safety=1.1, rule="global")
    c = ladder_thresholds(z1, None, z3, act, budget=9e-4, safety=1.1, rule="cross")
    want = 9e-4 / (1.1 * g["eps1"])
    assert g["r_global"] == pytest.approx(want, rel=1e-12) and c["r_cross"] >= g["r_global"] * 0.97
    # the cross threshold IS a grid point whose error exceeds the budget, and no grid point below it does
    bad = E * 1.1 > 9e-4
    assert bool(bad.any()) and c["r_cross"] == pytest.approx(float(R[bad].min()), rel=1e-12) and not bool(bad[R < c["r_cross"]].any())


def test_two_rungs_and_the_sigmoid_caps():
    z3 = _logits(seed=5)
    gen = torch.Generator().manual_seed(6)
    z1 = z3 + torch.randn(z3.shape, generator=gen) * 2.4e-3        # first rung: eps ~ 1.9e-3 (mean |N(0, s)| = 0.8 s)
    z2 = z3 + torch.randn(z3.shape, generator=gen) * 1.2e-3        # second rung: half of it
    res = ladder_thresholds(z1, z2, z3, ACT_SIGMOID, budget=9e-4, safety=1.1, rule="cross")
    assert res["eps1"] == pytest.approx(0.8 * 2.4e-3, rel=0.08) and res["eps2"] == pytest.approx(0.8 * 1.2e-3, rel=0.08)
    assert 0.3 < res["r"] < 0.6 and res["r"] <= res["r3"] <= 0.97 and res["r3_global"] == pytest.approx(2 * res["r_global"], rel=0.1)
    quiet = ladder_thresholds(z3 + (z1 - z3) * 1e-3, z3.clone(), z3, ACT_SIGMOID, budget=9e-4, safety=1.1, rule="cross")
    assert quiet["r"] == 0.97 and quiet["r3"] == 0.97 and math.isinf(quiet["r_cross"])     # a rung that never exceeds the budget is never left (capped: r < 1)
    loud = ladder_thresholds(z3 + (z1 - z3) * 100, z2, z3, ACT_SIGMOID, budget=9e-4, safety=1.1, rule="global")
    assert loud["r"] == 0.02 and loud["r3"] >= loud["r"]                                   # ... and one that always does is left at once
    with pytest.raises(Exception):
        ladder_thresholds(z1, None, z3, ACT_SIGMOID, 9e-4, 1.1, rule="other")


def test_relu_a_mostly_clipped_map_has_a_large_r():
    """The ReLU head's r = (positive outputs) / sum out: the same logit error weighs more the smaller the positive part of the map is (round 5's known limit)."""
    z3 = _logits(seed=7, spread=1.0)
    R, E = ladder_curve(z3 + 1e-3, z3, ACT_RELU)
    # shifts run from 97 % of the map positive (large outputs, small r) to 3 % (just above the kink, large r)
    assert bool((R[:, 1:] > R[:, :-1]).all()) and float(R[:, -1].min()) > 4 * float(R[:, 0].max())
