"""BASELINE config 1 (CartPole-swingup, SINDy model, MPPI 256 samples x 20 horizon) on the CPU
path: plumbing only.  SINDy inference is PARITY UNPINNED (pysindy is absent everywhere, see
oracle/sindy.py); these tests pin the restatement to itself: library layout, finite-difference
agreement of the non-strict Jacobian, and an end-to-end MPPI / iLQR closed loop that runs."""
import numpy as np

from helpers import make_system
from oracle.closed_loop import simulate
from oracle.costs import QuadCostOracle
from oracle.ilqr import ILQROracle
from oracle.mppi import MPPIOracle
from oracle.sindy import SINDyOracle, build_library


def _cartpole_like():
    system = make_system(4, 1, dt=0.05)
    feats = build_library(5, trig_freq=1, trig_interaction=True)
    rng = np.random.default_rng(0)
    Xi = np.zeros((4, len(feats)))
    Xi[:, :4] = np.eye(4)                                   # identity part: x' ~ x
    Xi[0, 2] = 0.05                                          # x += dt * xdot
    Xi[1, 3] = 0.05
    mask = rng.random(Xi.shape) < 0.08                       # sparse random remainder (STLSQ-like)
    Xi = Xi + mask * rng.normal(scale=0.02, size=Xi.shape)
    Xi[2, 4] = 0.1                                           # control enters the velocities
    Xi[3, 4] = -0.08
    return system, Xi


def test_library_layout():
    feats = build_library(5, trig_freq=1, trig_interaction=True)
    # 5 identity + 5 sin + 5 cos + 4 interaction kinds x C(5,2) pairs
    assert len(feats) == 5 + 5 + 5 + 4 * 10
    assert feats[0] == ("id", (0,), None) and feats[5] == ("sin", (0,), 1)
    assert feats[15] == ("xsin", (0, 1), 1) and feats[25] == ("xsin2", (0, 1), 1)
    assert len(build_library(3, trig_freq=2, poly_degree=3)) == 3 + 2 * 6 + 2 * 3


def test_nonstrict_jacobian_is_the_true_derivative():
    system, Xi = _cartpole_like()
    m = SINDyOracle(system, Xi, trig_freq=1, trig_interaction=True, strict_reference=False)
    rng = np.random.default_rng(1)
    s, c = rng.normal(size=(6, 4)), rng.normal(size=(6, 1))
    _, jx, ju = m.pred_diff_batch(s, c)
    h = 1e-6
    for j in range(4):
        d = np.zeros(4); d[j] = h
        fd = (m.pred_batch(s + d, c) - m.pred_batch(s - d, c)) / (2 * h)
        np.testing.assert_allclose(jx[:, :, j], fd, atol=1e-7)
    fd = (m.pred_batch(s, c + h) - m.pred_batch(s, c - h)) / (2 * h)
    np.testing.assert_allclose(ju[:, :, 0], fd, atol=1e-7)
    # strict mode: identity/sin/cos parts agree, interaction parts are doubled (reference quirk)
    ms = SINDyOracle(system, Xi, trig_freq=1, trig_interaction=True, strict_reference=True)
    _, jxs, _ = ms.pred_diff_batch(s, c)
    assert not np.allclose(jxs, jx)


def test_config1_mppi_plumbing_runs_and_is_deterministic():
    system, Xi = _cartpole_like()
    model = SINDyOracle(system, Xi, trig_freq=1, trig_interaction=True)
    cost = QuadCostOracle(np.diag([1.0, 10.0, 0.1, 0.1]), 0.01 * np.eye(1), np.eye(4), np.zeros(4))
    runs = []
    for _ in range(2):
        np.random.seed(0)
        ctl = MPPIOracle(model, cost, np.array([[-20.0, 20.0]]), horizon=20, num_path=256,
                         sigma=1.0, lmda=1.0)
        obs, ctrls = simulate(ctl, np.array([0.0, 0.2, 0.0, 0.0]), model, 5)
        runs.append((obs, ctrls))
    assert np.all(np.isfinite(runs[0][0])) and np.all(np.abs(runs[0][1]) <= 20.0 + 1e-12)
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    assert runs[0][0].shape == (6, 4) and runs[0][1].shape == (6, 1)


def test_ilqr_on_sindy_model_reduces_cost():
    system, Xi = _cartpole_like()
    model = SINDyOracle(system, Xi, trig_freq=1, trig_interaction=True, strict_reference=False)
    cost = QuadCostOracle(np.eye(4), 0.1 * np.eye(1), 5.0 * np.eye(4), np.zeros(4))
    ctl = ILQROracle(model, cost, 0.05, 15)
    x0 = np.array([0.3, -0.2, 0.1, 0.0])
    conv, states, ctrls, Ks, ks = ctl.solve(x0, np.zeros((15, 1)))
    zero = ILQROracle(model, cost, 0.05, 15)
    xs = np.zeros((16, 4)); xs[0] = x0
    for i in range(15):
        xs[i + 1] = model.pred(xs[i], np.zeros(1))
    assert ctl.final_obj < zero._objective(xs, np.zeros((15, 1)))
