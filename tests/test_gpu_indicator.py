"""Threshold / box terms in an MPPI controller's cost (ampc_set_indicator_costs): the reference's MPPI
charges whatever Cost the task holds, term by term (mppi.py:73-82; thresh_cost.py:27-38, 73-83).
Reference goldens (tests/golden/indmppi_*.npz, nu = 1) through every rollout kernel that can take their
shape, and the oracle for several controls, SINDy and wide linear models.  Needs MI355X."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import (check_weights, golden_params, hip_indicator_cost_from_golden, indicator_cost_from_golden,
                     make_system, rel_err)
from oracle import mlp as omlp
from oracle.costs import BoxCostOracle, QuadCostOracle, SumCostOracle, ThresholdCostOracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle

pytestmark = pytest.mark.gpu

NAMES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "indmppi_*.npz")))


def _mlp(system, p, precision="f64"):
    from autompc_amd import MLP
    m = MLP(system, n_hidden_layers=len(p["weights"]) - 1, nonlintype=p["activation"], precision=precision,
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    return m


@pytest.mark.parametrize("kernel", ["auto", "sixteen_rows", "run_time_shapes"])
@pytest.mark.parametrize("name", NAMES)
def test_mppi_with_indicator_terms_matches_reference_golden(name, kernel, monkeypatch):
    from autompc_amd import MPPI, Task
    if kernel == "sixteen_rows":
        monkeypatch.setenv("AMPC_QUAD", "0")        # (the small shapes default to the four-row rollout)
    if kernel == "run_time_shapes":
        monkeypatch.setenv("AMPC_STATIC", "0")
        monkeypatch.setenv("AMPC_JIT", "0")
    g = golden(name)
    nx = int(g["nx"])
    system = make_system(nx, 1)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    model = _mlp(system, p)
    task = Task(system)
    cost = hip_indicator_cost_from_golden(system, g)
    task.set_cost(cost)
    task.set_ctrl_bounds(np.array([g["bounds"][0]]), np.array([g["bounds"][1]]))
    assert MPPI.is_compatible(system, task, model)
    np.random.seed(int(g["np_seed"]))
    ctl = MPPI(system, task, model, horizon=int(g["H"]), num_path=int(g["N"]), sigma=float(g["sigma"]),
               lmda=float(g["lmda"]))
    np.testing.assert_array_equal(ctl.act_sequence, g["act0"])
    obs = np.random.default_rng(int(g["np_seed"]) + 99).uniform(-0.1, 0.1, size=nx)
    constate = np.concatenate([obs, np.zeros(1)])
    ref_model = MLPOracle(system, p)
    for r in range(int(g["n_runs"])):
        u, constate = ctl.run(constate, obs, return_details=True)
        assert rel_err(ctl.last_costs, g["costs_%d" % r]) < 1e-9
        assert rel_err(ctl.act_sequence, g["act_%d" % r]) < 1e-8
        assert rel_err(u, g["u_%d" % r]) < 1e-8
        obs = ref_model.pred(obs, g["u_%d" % r])


def _costs(system, no, nu, rng, dense):
    from autompc_amd import BoxThresholdCost, QuadCost, ThresholdCost
    if dense:
        W = rng.normal(size=(no, no))
        Q, F = W @ W.T / no + 0.1 * rng.normal(size=(no, no)), np.diag(rng.uniform(0.5, 2, size=no))
        R = np.diag(rng.uniform(0.01, 0.1, size=nu)) + 0.002
    else:
        Q, F = np.diag(rng.uniform(0.5, 2, size=no)), np.diag(rng.uniform(0.5, 2, size=no))
        R = np.diag(rng.uniform(0.01, 0.1, size=nu))
    goal = rng.normal(scale=0.05, size=no)
    lo, hi = 1, max(2, no - 1)
    limits = np.stack([goal - rng.uniform(0.05, 0.3, size=no), goal + rng.uniform(0.05, 0.3, size=no)], axis=1)
    limits[0, 0], limits[no - 1, 1] = -np.inf, np.inf
    thr = 0.12
    hip = (QuadCost(system, Q, R, F, goal=goal) + ThresholdCost(system, goal, [lo, hi], thr)
           + BoxThresholdCost(system, limits) + ThresholdCost(system, goal + 0.02, [0, no], 2 * thr))
    orc = SumCostOracle([QuadCostOracle(Q, R, F, goal), ThresholdCostOracle(goal, [lo, hi], thr), BoxCostOracle(limits),
                         ThresholdCostOracle(goal + 0.02, [0, no], 2 * thr)])
    return hip, orc


def _compare(system, model, omodel, hip_cost, orc_cost, nu, N, H, x, no, tol=1e-9, runs=2, **kw):
    from autompc_amd import MPPI, Task
    task = Task(system)
    task.set_cost(hip_cost)
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    np.random.seed(3)
    orc = MPPIOracle(omodel, orc_cost, np.tile([-1.0, 1.0], (nu, 1)), horizon=H, num_path=N, sigma=0.7, lmda=0.9, **kw)
    np.random.seed(3)
    ctl = MPPI(system, task, model, horizon=H, num_path=N, sigma=0.7, lmda=0.9, **kw)
    cs = np.concatenate([x, np.zeros(nu)])
    levels = set()
    for _ in range(runs):
        st = np.random.get_state()
        uo, _ = orc.run(cs, x[:no])
        np.random.set_state(st)
        uh, _ = ctl.run(cs, x[:no], return_details=True)
        assert rel_err(ctl.last_costs, orc.last_costs) < tol
        assert rel_err(ctl.act_sequence, orc.act_sequence) < tol * 10 and rel_err(uh, uo) < tol * 10
        ctl.act_sequence = orc.act_sequence
        levels |= set(np.round(orc.last_costs).astype(int).tolist())
    assert len(levels) > 2          # the indicator terms tell samples apart


@pytest.mark.parametrize("case", [
    # nx, nu, hidden, act, N, H, dense cost, per-particle terminal
    (17, 6, [256, 256], "relu", 512, 12, False, False),       # c3 shape, static kernels, diagonal cost path
    (17, 6, [256, 256], "tanh", 300, 9, True, True),          # dense cost path
    (5, 2, [64, 48], "tanh", 96, 10, True, False),            # four-row rollout (small problem), dense
    (3, 2, [64, 64], "relu", 64, 14, False, False),           # four-row rollout, diagonal
    (40, 3, [64], "tanh", 200, 8, False, False),              # wide states
])
def test_several_controls_vs_oracle(case):
    nx, nu, hidden, act, N, H, dense, ppt = case
    p = omlp.random_params(nx, nu, hidden, act, seed=nx + 2)
    system = make_system(nx, nu)
    rng = np.random.default_rng(nx)
    hip, orc = _costs(system, nx, nu, rng, dense)
    x = rng.uniform(-0.1, 0.1, size=nx)
    _compare(system, _mlp(system, p), MLPOracle(system, p), hip, orc, nu, N, H, x, nx, per_particle_terminal=ppt)


@pytest.mark.parametrize("fp", ["1", "0"])
def test_sindy_rollouts_vs_oracle(fp, monkeypatch):
    """Both SINDy rollout kernels (features spread over lanes / one thread per sample)."""
    from autompc_amd import SINDy
    from oracle.sindy import SINDyOracle
    monkeypatch.setenv("AMPC_SINDY_FP", fp)
    system = make_system(4, 1, dt=0.05)
    m = SINDy(system, trig_basis=True, trig_freq=1, trig_interaction=True, poly_basis=False, poly_degree=1,
              time_mode="discrete", strict_reference=True)
    rng = np.random.default_rng(0)
    Xi = np.zeros((4, m.coefficients.shape[1]))
    Xi[:, :4] = np.eye(4)
    Xi[0, 2] = Xi[1, 3] = 0.05
    Xi += (rng.random(Xi.shape) < 0.1) * rng.normal(scale=0.02, size=Xi.shape)
    Xi[2, 4], Xi[3, 4] = 0.1, -0.08
    m.set_coefficients(Xi)
    om = SINDyOracle(system, Xi, trig_freq=1, trig_interaction=True, poly_degree=1, time_mode="discrete",
                     strict_reference=True)
    hip, orc = _costs(system, 4, 1, np.random.default_rng(8), dense=fp == "0")
    _compare(system, m, om, hip, orc, 1, 256, 20, np.array([0.0, 0.2, 0.0, 0.0]), 4)


def test_wide_linear_model_vs_oracle():
    from autompc_amd.sysid.model import Model
    ns, nu, no, N, H = 91, 6, 18, 150, 9
    rng = np.random.default_rng(ns)
    A = 0.9 * np.linalg.qr(rng.normal(size=(ns, ns)))[0] + 0.02 * rng.normal(size=(ns, ns)) / np.sqrt(ns)
    Bm = rng.normal(scale=0.3, size=(ns, nu))
    system = make_system(no, nu)

    class Lin(Model):
        def __init__(self):
            super().__init__(system)
            self.precision, self.device, self._h = "f64", 0, None

        @property
        def state_dim(self):
            return ns

        def stage_into(self, h):
            h.set_linear(A, Bm)

        def update_state(self, state, ctrl, obs):
            return np.asarray(state).copy()

        def traj_to_state(self, traj):
            raise NotImplementedError

        def pred(self, s, u):
            return A @ s + Bm @ u

        def pred_batch(self, s, u):
            return s @ A.T + u @ Bm.T
    hip, orc = _costs(system, no, nu, rng, dense=True)
    model = Lin()
    _compare(system, model, model, hip, orc, nu, N, H, rng.uniform(-0.1, 0.1, size=ns), no)


def test_bare_box_cost_and_ilqr_refusal():
    """A cost without any quadratic term stages a zero block; iLQR takes sums of quadratics only."""
    from autompc_amd import BoxThresholdCost, IterativeLQR, MPPI, Task, _lib
    nx, nu = 3, 2
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [64, 64], "tanh", seed=4)
    model = _mlp(system, p)
    limits = np.array([[-0.05, 0.05], [-np.inf, 0.1], [-0.2, np.inf]])
    task = Task(system)
    task.set_cost(BoxThresholdCost(system, limits))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    assert MPPI.is_compatible(system, task, model) and not IterativeLQR.is_compatible(system, task, model)
    np.random.seed(1)
    orc = MPPIOracle(MLPOracle(system, p), SumCostOracle([BoxCostOracle(limits)]), np.tile([-1.0, 1.0], (nu, 1)),
                     horizon=8, num_path=64, sigma=1.0, lmda=1.0)
    np.random.seed(1)
    ctl = MPPI(system, task, model, horizon=8, num_path=64, sigma=1.0, lmda=1.0)
    x = np.array([0.01, -0.02, 0.03])
    cs = np.concatenate([x, np.zeros(nu)])
    st = np.random.get_state()
    uo, _ = orc.run(cs, x)
    np.random.set_state(st)
    uh, _ = ctl.run(cs, x, return_details=True)
    assert rel_err(ctl.last_costs, orc.last_costs) < 1e-9 and rel_err(uh, uo) < 1e-8
    # the C ABI refuses an iLQR plan on a handle whose cost has indicator terms
    h = _lib.Handle(0, "f64")
    model.stage_into(h)
    h.set_quad_costs(np.eye(nx), np.eye(nu), np.eye(nx), np.zeros(nx))
    h.set_indicator_costs((np.array([2], dtype=np.int32), np.concatenate([limits[:, 0], limits[:, 1]])))
    with pytest.raises(RuntimeError, match="indicator"):
        _lib.IlqrPlan(h, 1, 5, 0.05)
    h.set_indicator_costs(None)
    _lib.IlqrPlan(h, 1, 5, 0.05).close()
    h.close()
