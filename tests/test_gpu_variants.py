"""Every kernel instantiation the dispatcher can pick -- workgroup shapes (W waves, NT tiles per
wave) and tile heights (MT) -- checked against the oracle.  The defaults only exercise a few of
them; the override AMPC_MT forces the rest (needs MI355X)."""
import numpy as np
import pytest

from helpers import make_system, rel_err
from oracle import mlp as omlp
from oracle.costs import QuadCostOracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle

pytestmark = pytest.mark.gpu

# hidden sizes -> padded width -> (W, NT) by default:  64->(4,1) 128->(8,1) 192->(4,3) 256->(8,2)
SHAPES = [([40, 64], None), ([100, 128], None), ([150, 192], None), ([256, 200], None)]


def _handle(p, nx, nu, act, precision):
    from autompc_amd import _lib
    h = _lib.Handle(0, precision)
    h.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"],
              p["dy_std"])
    return h


@pytest.mark.parametrize("precision,tol", [("f64", 1e-10), ("f32", 1e-4)])
@pytest.mark.parametrize("mt", ["1", "2", "4"])
@pytest.mark.parametrize("hidden,waves", SHAPES)
def test_forward_and_jacobian_all_instantiations(monkeypatch, hidden, waves, mt, precision, tol):
    # (a forced tile height that does not fit the 160 KB LDS falls back to the largest that does)
    monkeypatch.setenv("AMPC_MT", mt)
    nx, nu = 11, 3
    p = omlp.random_params(nx, nu, hidden, "tanh", seed=sum(hidden))
    rng = np.random.default_rng(1)
    p["xu_means"], p["xu_std"] = rng.normal(scale=0.2, size=nx + nu), rng.uniform(0.5, 2, size=nx + nu)
    p["dy_means"], p["dy_std"] = rng.normal(scale=0.05, size=nx), rng.uniform(0.05, 0.3, size=nx)
    h = _handle(p, nx, nu, "tanh", precision)
    s, c = rng.normal(size=(150, nx)), rng.normal(size=(150, nu))
    assert rel_err(h.pred_batch(s, c), omlp.pred_batch(p, s, c)) < tol
    o, jx, ju = h.pred_diff_batch(s[:40], c[:40])
    eo, ejx, eju = omlp.pred_diff_batch(p, s[:40], c[:40])
    assert rel_err(o, eo) < tol and rel_err(jx, ejx) < 10 * tol and rel_err(ju, eju) < 10 * tol


@pytest.mark.parametrize("precision,tol", [("f64", 1e-9), ("f32", 1e-4)])
@pytest.mark.parametrize("mt", ["1", "2", "4"])
@pytest.mark.parametrize("hidden,waves", SHAPES)
def test_mppi_solve_all_instantiations(monkeypatch, hidden, waves, mt, precision, tol):
    # (a forced tile height that does not fit the 160 KB LDS falls back to the largest that does)
    from autompc_amd import _lib
    monkeypatch.setenv("AMPC_MT", mt)
    nx, nu, N, H = 9, 4, 203, 7          # N is not a multiple of any tile height
    p = omlp.random_params(nx, nu, hidden, "relu", seed=7 + sum(hidden))
    rng = np.random.default_rng(2)
    Q, R, F = rng.normal(size=(nx, nx)), np.diag(rng.uniform(0.01, 0.1, size=nu)), np.eye(nx)
    goal = rng.normal(scale=0.1, size=nx)                 # dense, non-symmetric Q: dense-cost path
    h = _handle(p, nx, nu, "relu", precision)
    h.set_quad_costs(Q, R, F, goal)
    lo, hi = -rng.uniform(0.5, 1.5, size=nu), rng.uniform(0.5, 1.5, size=nu)
    h.set_ctrl_bounds(lo, hi)
    plan = _lib.MppiPlan(h, [N], [H], [0.8], [0.6], term_mode=_lib.TERM_PER_PARTICLE)
    assert plan.info()["samples_per_wg"] == 16 * int(mt)
    x0 = rng.uniform(-0.1, 0.1, size=nx)
    act = rng.normal(scale=0.5, size=(H, nu))
    eps = rng.normal(scale=np.sqrt(0.8), size=(N, H, nu))
    plan.upload(x0, act, eps)
    plan.solve()
    a, u, c, e = plan.download(costs=True, eps_out=True)
    orc = MPPIOracle(MLPOracle(make_system(nx, nu), p), QuadCostOracle(Q, R, F, goal),
                     np.stack([lo, hi], axis=1), horizon=H, num_path=N, sigma=0.8, lmda=0.6,
                     per_particle_terminal=True)
    orc.act_sequence = act.copy()
    uo, _ = orc.run(np.concatenate([x0, np.zeros(nu)]), x0, eps_nhu=eps)
    assert rel_err(c, orc.last_costs) < tol
    assert rel_err(e.reshape(H, N, nu), orc.last_eps) < max(tol, 1e-12)
    assert rel_err(a.reshape(H, nu), orc.act_sequence) < 10 * tol and rel_err(u[0], uo) < 10 * tol


@pytest.mark.parametrize("nx,nu,H", [(1, 1, 2), (32, 16, 5), (17, 6, 30), (3, 2, 64)])
def test_mppi_dimension_extremes(nx, nu, H):
    """Smallest / largest supported state and control widths, shortest horizon, a long one."""
    from autompc_amd import _lib
    N = 50
    p = omlp.random_params(nx, nu, [64, 64], "sigmoid", seed=nx)
    rng = np.random.default_rng(nx + nu)
    Q, R, F = np.diag(rng.uniform(0.5, 2, nx)), np.diag(rng.uniform(0.01, 0.1, nu)), np.eye(nx)
    h = _handle(p, nx, nu, "sigmoid", "f64")
    h.set_quad_costs(Q, R, F, np.zeros(nx))
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
    x0, act = rng.uniform(-0.1, 0.1, size=nx), rng.normal(size=(H, nu))
    eps = rng.normal(size=(N, H, nu))
    plan.upload(x0, act, eps)
    plan.solve()
    a, u, c, _ = plan.download(costs=True)
    orc = MPPIOracle(MLPOracle(make_system(nx, nu), p), QuadCostOracle(Q, R, F, np.zeros(nx)),
                     np.tile([-1.0, 1.0], (nu, 1)), horizon=H, num_path=N)
    orc.act_sequence = act.copy()
    uo, _ = orc.run(np.concatenate([x0, np.zeros(nu)]), x0, eps_nhu=eps)
    assert rel_err(c, orc.last_costs) < 1e-9 and rel_err(u[0], uo) < 1e-8


def test_unsupported_shapes_fail_loudly():
    from autompc_amd import _lib
    h = _lib.Handle(0, "f64")
    for nx, hidden in ((65, [64]), (4, [300]), (70, [256, 256])):   # more than 64 states / 256 units
        p = omlp.random_params(nx, 1, hidden, "relu", seed=0)
        with pytest.raises(_lib.AmpcError):
            h.set_mlp(nx, 1, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"],
                      p["dy_std"])
    h.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("nx,nu,hidden,act", [(33, 1, [64], "relu"), (40, 3, [64, 64], "tanh"),
                                              (64, 2, [48, 64], "relu"), (50, 6, [32], "selu"),
                                              # hidden layers wider than 64 (every (waves, column-tile) shape)
                                              (33, 2, [128], "relu"), (41, 6, [256, 256], "relu"),
                                              (48, 4, [192, 160], "tanh"), (64, 8, [100, 128, 90], "sigmoid"),
                                              (37, 3, [256, 200, 256, 130], "selu")])
def test_wide_state_networks_match_the_oracle(nx, nu, hidden, act, precision):
    """33..64 model states (three or four output tiles), hidden layers of any width the reference's
    configuration space allows (mlp.py:113-122): prediction, Jacobians and an MPPI solve."""
    from autompc_amd import _lib
    p = omlp.random_params(nx, nu, hidden, act, seed=nx)
    h = _handle(p, nx, nu, act, precision)
    rng = np.random.default_rng(nx)
    n = 77
    X, U = rng.normal(size=(n, nx)), rng.normal(size=(n, nu))
    ref = omlp.pred_batch(p, X, U)
    tol = 1e-12 if precision == "f64" else 2e-5
    assert rel_err(h.pred_batch(X, U), ref) < tol
    o, jx, ju = h.pred_diff_batch(X, U)
    _, rjx, rju = omlp.pred_diff_batch(p, X, U)[:3]
    assert rel_err(o, ref) < tol and rel_err(jx, rjx) < 10 * tol and rel_err(ju, rju) < 10 * tol
    if precision == "f32":
        h.close()
        return
    N, H = 100, 7
    Q, R, F = np.eye(nx), 0.1 * np.eye(nu), 2 * np.eye(nx)
    h.set_quad_costs(Q, R, F, np.zeros(nx))
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
    x0 = rng.normal(size=nx)
    eps = rng.normal(size=(N, H, nu))
    act0 = rng.uniform(-0.3, 0.3, size=(H, nu))
    plan.upload(x0, act0, eps)
    plan.solve()
    a, u, c, _ = plan.download(costs=True)
    orc = MPPIOracle(MLPOracle(make_system(nx, nu), p), QuadCostOracle(Q, R, F, np.zeros(nx)),
                     np.tile([-1.0, 1.0], (nu, 1)), horizon=H, num_path=N)
    orc.act_sequence = act0.copy()
    uo, _ = orc.run(np.concatenate([x0, np.zeros(nu)]), x0, eps_nhu=eps)
    assert rel_err(c, orc.last_costs) < 1e-9 and rel_err(u[0], uo) < 1e-8
    plan.close()
    for tile_rows in (32, 64):                         # taller tiles of the same solve
        plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
        try:
            plan.set_geometry(tile_rows, 0)
        except _lib.AmpcError:                          # (that height does not fit LDS for this model)
            plan.close()
            continue
        plan.upload(x0, act0, eps)
        plan.solve()
        _, u2, c2, _ = plan.download(costs=True)
        assert rel_err(c2, orc.last_costs) < 1e-9 and rel_err(u2[0], uo) < 1e-8
        plan.close()
    if nx + nu <= 63:                                  # (the Riccati workspace of a wide model lives in one wave)
        from oracle.ilqr import ILQROracle
        Hh, dt = 8, 0.05
        try:
            iplan = _lib.IlqrPlan(h, 2, Hh, dt)
        except _lib.AmpcError as e:                    # (f64 Riccati workspace of ~50 states: > 160 KB LDS)
            assert "LDS" in str(e)
            h.close()
            return
        xs = rng.uniform(-0.2, 0.2, size=(2, nx))
        out = iplan.solve(xs, np.zeros((2, Hh, nu)), max_iter=6)
        for b in range(2):
            io = ILQROracle(MLPOracle(make_system(nx, nu, dt=dt), p), QuadCostOracle(Q, R, F, np.zeros(nx)), dt, Hh,
                            max_iter=6)
            conv, st, ct, _, _ = io.solve(xs[b], np.zeros((Hh, nu)))
            assert int(out["iters"][b]) == io.n_iter and out["status"][b] == 0
            assert rel_err(out["states"][b], st) < 1e-7 and rel_err(out["ctrls"][b], ct) < 1e-6
        iplan.close()
    h.close()
