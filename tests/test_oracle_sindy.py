"""BASELINE config 1 (CartPole-swingup, SINDy model, MPPI 256 samples x 20 horizon) on the CPU
path.  oracle/sindy.py is pinned by tests/golden/sindy_*.npz: outputs of the reference's own
SINDy.pred_batch / pred_diff_batch / compute_gradient (sindy.py:173-244) with basis functions
built by its own train() (sindy.py:134-152, basis_funcs.py:8-126) on a stand-in for the absent
pysindy package (gen_golden.py: only CustomLibrary's feature enumeration is restated there),
plus the reference's MPPI (config 1: 256 x 20) and iLQR on that model."""
import numpy as np
import pytest

from conftest import golden
from helpers import make_system, rel_err
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


SINDY_GOLDENS = ["sindy_c1_trig", "sindy_poly3_trig2_cont", "sindy_poly4_disc", "sindy_identity_cont",
                 "sindy_cross3", "sindy_cross4_trig_cont"]


def oracle_from_golden(g, strict=True):
    system = make_system(int(g["nx"]), int(g["nu"]), dt=float(g["dt"]))
    return system, SINDyOracle(system, g["Xi"], trig_freq=int(g["trig_freq"]),
                               trig_interaction=bool(g["trig_interaction"]),
                               poly_degree=int(g["poly_degree"]), time_mode=str(g["time_mode"]),
                               strict_reference=strict,
                               poly_cross_terms=bool(g["poly_cross_terms"]) if "poly_cross_terms" in g.files else False)


@pytest.mark.parametrize("name", SINDY_GOLDENS)
def test_sindy_oracle_matches_reference(name):
    """Prediction and the name-lookup Jacobian incl. its two quirks (polynomial gradient without
    the exponent factor, interaction gradients counted twice) against the reference's outputs."""
    g = golden(name)
    _, m = oracle_from_golden(g)
    assert m.Xi.shape[1] == len(g["feature_names"])
    assert rel_err(m.pred_batch(g["states"], g["ctrls"]), g["pred_batch"]) < 1e-13
    o, jx, ju = m.pred_diff_batch(g["states"], g["ctrls"])
    assert rel_err(o, g["diff_pred"]) < 1e-13
    assert rel_err(jx, g["diff_jx"]) < 1e-13 and rel_err(ju, g["diff_ju"]) < 1e-13
    assert rel_err(m.pred(g["states"][0], g["ctrls"][0]), g["pred0"]) < 1e-13
    o0, jx0, ju0 = m.pred_diff(g["states"][0], g["ctrls"][0])
    assert rel_err(jx0, g["diff0_jx"]) < 1e-13 and rel_err(ju0, g["diff0_ju"]) < 1e-13
    if int(g["poly_degree"]) > 1 or bool(g["trig_interaction"]):
        # the quirks are real: the true derivative differs from what the reference returns
        _, mt = oracle_from_golden(g, strict=False)
        assert rel_err(mt.pred_diff_batch(g["states"], g["ctrls"])[1], g["diff_jx"]) > 1e-3


def test_feature_order_matches_reference_names():
    g = golden("sindy_poly3_trig2_cont")
    feats = build_library(5, trig_freq=2, trig_interaction=True, poly_degree=3)
    var = ["x0", "x1", "x2", "u0", "u1"]
    fmt = {"id": lambda c, p: var[c[0]], "sin": lambda c, p: "sin(%d %s)" % (p, var[c[0]]),
           "cos": lambda c, p: "cos(%d %s)" % (p, var[c[0]]),
           "xsin": lambda c, p: "%s sin(%d %s)" % (var[c[0]], p, var[c[1]]),
           "xsin2": lambda c, p: "%s sin(%d %s)" % (var[c[1]], p, var[c[0]]),
           "xcos": lambda c, p: "%s cos(%d %s)" % (var[c[0]], p, var[c[1]]),
           "xcos2": lambda c, p: "%s cos(%d %s)" % (var[c[1]], p, var[c[0]]),
           "pow": lambda c, p: "%s**%d" % (var[c[0]], p)}
    assert [fmt[k](c, p) for k, c, p in feats] == [str(n) for n in g["feature_names"]]


def test_config1_mppi_matches_reference():
    """BASELINE config 1: the reference's MPPI (256 samples x 20 horizon) on the SINDy model."""
    g = golden("sindy_c1_trig")
    _, model = oracle_from_golden(g)
    np.random.seed(int(g["np_seed"]))
    ctl = MPPIOracle(model, QuadCostOracle(g["Q"], g["R"], g["F"], np.zeros(4)), np.array([g["bounds"]]),
                     horizon=int(g["H"]), num_path=int(g["N"]), sigma=float(g["sigma"]),
                     lmda=float(g["lmda"]))
    np.testing.assert_array_equal(ctl.act_sequence, g["mppi_act0"])
    obs = np.array([0.0, 0.2, 0.0, 0.0])
    cs = np.concatenate([obs, np.zeros(1)])
    for r in range(3):
        np.testing.assert_allclose(obs, g["mppi_x0_%d" % r], rtol=1e-10, atol=1e-13)
        u, cs = ctl.run(cs, obs)
        assert rel_err(ctl.last_costs, g["mppi_costs_%d" % r]) < 1e-10
        assert rel_err(ctl.act_sequence, g["mppi_act_%d" % r]) < 1e-9
        assert rel_err(u, g["mppi_u_%d" % r]) < 1e-9
        obs = model.pred(obs, u)


def test_ilqr_on_sindy_matches_reference():
    g = golden("sindy_c1_trig")
    _, model = oracle_from_golden(g)
    H = int(g["ilqr_H"])
    orc = ILQROracle(model, QuadCostOracle(g["Q"], g["R"], g["F"], np.zeros(4)), 0.05, H)
    conv, st, ct, Ks, ks = orc.solve(g["ilqr_x0"], np.zeros((H, 1)))
    assert conv == bool(g["ilqr_converged"])
    assert rel_err(st, g["ilqr_states"]) < 1e-7 and rel_err(ct, g["ilqr_ctrls"]) < 1e-7
    assert rel_err(Ks, g["ilqr_Ks"]) < 1e-6


def test_cross_term_library_matches_reference_names():
    """Polynomial cross terms (basis_funcs.py:27-93): exponent tuples in the reference's order and
    the features pysindy's enumeration makes of them, against the reference's own feature names."""
    from oracle.sindy import cross_term_exponents
    assert cross_term_exponents(2) == [(1, 1)]
    assert cross_term_exponents(3) == [(1, 2), (2, 1), (1, 1, 1)]
    assert cross_term_exponents(4) == [(1, 3), (2, 2), (3, 1), (1, 1, 2), (1, 2, 1), (2, 1, 1), (1, 1, 1, 1)]
    g = golden("sindy_cross3")
    feats = build_library(5, poly_degree=3, poly_cross_terms=True)
    var = ["x0", "x1", "x2", "u0", "u1"]
    names = []
    for kind, c, p in feats:
        if kind == "id":
            names.append(var[c[0]])
        elif kind == "pow":
            names.append("%s**%d" % (var[c[0]], p))
        else:
            names.append("".join("%s^%d " % (var[j], e) for j, e in zip(c, p)))
    assert names == [str(n) for n in g["feature_names"]]
    # the product library (what the device is handed) enumerates the same features
    from autompc_amd.sysid.sindy import K_MONO, build_library as product_library
    kind, a0, a1, par, pv, pe = product_library(5, poly_degree=3, poly_cross_terms=True)
    assert len(kind) == len(feats)
    for k, (okind, c, p) in enumerate(feats):
        if okind == "mono":
            assert kind[k] == K_MONO
            assert tuple(pv[a0[k]:a0[k] + a1[k]]) == c and tuple(pe[a0[k]:a0[k] + a1[k]]) == p
