"""Pin the CPU oracle to the real reference (golden vectors made by
tests/golden/gen_golden.py from /root/reference).  CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import (check_weights, cost_from_golden, golden_params, make_system, rel_err)
from oracle import mlp as omlp
from oracle.analytic import CubicIntegrator
from oracle.closed_loop import eval_cfg_episode, simulate
from oracle.costs import QuadCostOracle
from oracle.ilqr import ILQROracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle

TIGHT = 1e-11


def _names(prefix):
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


@pytest.mark.parametrize("name", _names("mlp_"))
def test_mlp_matches_reference(name):
    g = golden(name)
    nx, nu = int(g["nx"]), int(g["nu"])
    p = golden_params(nx, nu, g["hidden"], g["activation"], g["seed"])
    check_weights(p, g)
    out = omlp.pred_batch(p, g["states"], g["ctrls"])
    assert rel_err(out, g["pred_batch"]) < TIGHT
    o2, jx, ju = omlp.pred_diff_batch(p, g["states"], g["ctrls"])
    assert rel_err(o2, g["diff_pred"]) < TIGHT
    assert rel_err(jx, g["diff_jx"]) < 1e-10
    assert rel_err(ju, g["diff_ju"]) < 1e-10
    m = MLPOracle(make_system(nx, nu), p)
    assert rel_err(m.pred(g["states"][0], g["ctrls"][0]), g["pred0"]) < TIGHT
    o0, a0, b0 = m.pred_diff(g["states"][0], g["ctrls"][0])
    assert rel_err(o0, g["diff0_pred"]) < TIGHT
    assert rel_err(a0, g["diff0_jx"]) < 1e-10 and rel_err(b0, g["diff0_ju"]) < 1e-10


def test_quadcost_matches_reference():
    g = golden("cost_quad")
    c = QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"])
    obs, ctrl = g["obs"], g["ctrl"]
    assert abs(c.eval_obs_cost(obs) - g["obs_cost"]) < 1e-12
    assert abs(c.eval_ctrl_cost(ctrl) - g["ctrl_cost"]) < 1e-12
    assert abs(c.eval_term_obs_cost(obs) - g["term_cost"]) < 1e-12
    for got, key in zip(c.eval_obs_cost_hess(obs), ("obs_c", "obs_j", "obs_h")):
        np.testing.assert_allclose(got, g[key], rtol=1e-13, atol=1e-13)
    for got, key in zip(c.eval_ctrl_cost_hess(ctrl), ("ctrl_c", "ctrl_j", "ctrl_h")):
        np.testing.assert_allclose(got, g[key], rtol=1e-13, atol=1e-13)
    # terminal grad/hess ignore the goal in the reference (cost.py:195, 208-211)
    for got, key in zip(c.eval_term_obs_cost_hess(obs), ("term_c", "term_j", "term_h")):
        np.testing.assert_allclose(got, g[key], rtol=1e-13, atol=1e-13)
    assert abs(c.traj_cost(g["traj_obs"], g["traj_ctrls"]) - g["traj_cost"]) < 1e-11


def _mppi_from_golden(g, strict=False):
    nx = int(g["nx"])
    system = make_system(nx, 1)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], bool(g["plain_norm"]))
    check_weights(p, g)
    model = MLPOracle(system, p)
    bounds = np.array([[g["bounds"][0], g["bounds"][1]]])
    np.random.seed(int(g["np_seed"]))
    ctl = MPPIOracle(model, cost_from_golden(g), bounds, horizon=int(g["H"]),
                     num_path=int(g["N"]), sigma=float(g["sigma"]), lmda=float(g["lmda"]),
                     strict_reference=strict)
    return model, ctl


@pytest.mark.parametrize("name", _names("mppi_"))
def test_mppi_matches_reference(name):
    g = golden(name)
    model, ctl = _mppi_from_golden(g)
    np.testing.assert_allclose(ctl.act_sequence, g["act0"], rtol=0, atol=0)
    nx = int(g["nx"])
    obs = np.random.default_rng(int(g["np_seed"]) + 99).uniform(-0.1, 0.1, size=nx)
    constate = np.concatenate([obs, np.zeros(1)])
    for r in range(int(g["n_runs"])):
        if r == 3:
            ctl.reset()
            np.testing.assert_allclose(ctl.act_sequence, g["act_reset"], rtol=0, atol=0)
        np.testing.assert_allclose(obs, g["x0_%d" % r], rtol=1e-10, atol=1e-12)
        u, constate = ctl.run(constate, obs)
        assert rel_err(ctl.last_costs, g["costs_%d" % r]) < 1e-10
        assert rel_err(ctl.last_eps[:, ::16, :], g["eps_sub_%d" % r]) < 1e-12
        assert rel_err(ctl.act_sequence, g["act_%d" % r]) < 1e-9
        assert rel_err(u, g["u_%d" % r]) < 1e-9
        assert rel_err(constate, g["newstate_%d" % r]) < 1e-9
        obs = model.pred(obs, u)


@pytest.mark.parametrize("name", _names("indmppi_"))
def test_mppi_with_threshold_and_box_terms_matches_reference(name):
    """MPPI on QuadCost + ThresholdCost, a bare BoxThresholdCost and quad + threshold + box
    (mppi.py:73-82 charges the task's Cost term by term; thresh_cost.py:27-38, 73-83)."""
    from helpers import indicator_cost_from_golden
    g = golden(name)
    nx = int(g["nx"])
    system = make_system(nx, 1)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    model = MLPOracle(system, p)
    for strict in (False, True):
        np.random.seed(int(g["np_seed"]))
        ctl = MPPIOracle(model, indicator_cost_from_golden(g), np.array([[g["bounds"][0], g["bounds"][1]]]),
                         horizon=int(g["H"]), num_path=int(g["N"]), sigma=float(g["sigma"]), lmda=float(g["lmda"]),
                         strict_reference=strict)
        np.testing.assert_array_equal(ctl.act_sequence, g["act0"])
        obs = np.random.default_rng(int(g["np_seed"]) + 99).uniform(-0.1, 0.1, size=nx)
        constate = np.concatenate([obs, np.zeros(1)])
        for r in range(int(g["n_runs"])):
            np.testing.assert_allclose(obs, g["x0_%d" % r], rtol=1e-10, atol=1e-12)
            u, constate = ctl.run(constate, obs)
            assert rel_err(ctl.last_costs, g["costs_%d" % r]) < 1e-10
            assert rel_err(ctl.act_sequence, g["act_%d" % r]) < 1e-9
            assert rel_err(u, g["u_%d" % r]) < 1e-9
            obs = model.pred(obs, u)


def test_mppi_strict_loop_equals_vectorised():
    g = golden("mppi_clip_asym")
    _, a = _mppi_from_golden(g, strict=True)
    _, b = _mppi_from_golden(g, strict=False)
    x0 = np.array([0.05, -0.02])
    cs = np.concatenate([x0, np.zeros(1)])
    np.random.seed(7)
    ua, _ = a.run(cs, x0)
    np.random.seed(7)
    b.act_sequence = g["act0"].copy()
    a_costs = a.last_costs
    ub, _ = b.run(cs, x0)
    assert rel_err(b.last_costs, a_costs) < 1e-12 and rel_err(ub, ua) < 1e-12


def _ilqr_from_golden(g):
    nx, nu = int(g["nx"]), int(g["nu"])
    system = make_system(nx, nu, dt=float(g["dt"]))
    if str(g["kind"]) == "cubic":
        model = CubicIntegrator(system)
    else:
        p = golden_params(nx, nu, g["hidden"], g["activation"], g["mlp_seed"], bool(g["plain_norm"]))
        check_weights(p, g)
        model = MLPOracle(system, p)
    ub = None
    if bool(g["bounded"]):
        ub = (np.full(nu, g["bounds"][0]), np.full(nu, g["bounds"][1]))
    return ILQROracle(model, cost_from_golden(g), float(g["dt"]), int(g["H"]), ubounds=ub)


@pytest.mark.parametrize("name", _names("ilqr_"))
def test_ilqr_matches_reference(name):
    g = golden(name)
    ctl = _ilqr_from_golden(g)
    nu = int(g["nu"])
    conv, states, ctrls, Ks, ks = ctl.solve(g["x0"], np.zeros((int(g["H"]), nu)))
    assert conv == bool(g["converged"])
    assert sum(1 for t in ctl.trace if t[3]) == int(g["n_refresh"])
    # iLQR amplifies rounding through up to 50 Riccati sweeps; the non-converged tanh
    # case is chaotic in the last digits, so it gets a looser bound.
    tol = 1e-6 if conv else 1e-4
    assert rel_err(states, g["states"]) < tol
    assert rel_err(ctrls, g["ctrls"]) < tol
    assert rel_err(Ks, g["Ks"]) < tol * 10
    assert rel_err(ks, g["ks"]) < tol * 10 or np.max(np.abs(ks - g["ks"])) < 1e-9
    u, newstate = ctl.run(np.concatenate([g["x0"], np.zeros(nu)]), g["x0"])
    assert rel_err(u, g["u"]) < tol and rel_err(newstate, g["newstate"]) < tol


def test_closed_loop_mppi_matches_reference():
    g = golden("loop_mppi")
    nx = int(g["nx"])
    system = make_system(nx, 1)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    model = MLPOracle(system, p)
    cost = cost_from_golden(g)
    np.random.seed(int(g["np_seed"]))
    ctl = MPPIOracle(model, cost, np.array([g["bounds"]]), horizon=int(g["H"]), num_path=int(g["N"]),
                     sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    obs, ctrls = simulate(ctl, g["init"], model, 20)
    assert rel_err(obs, g["obs"]) < 1e-8 and rel_err(ctrls, g["ctrls"]) < 1e-8
    assert abs(cost.traj_cost(obs, ctrls) - g["score"]) < 1e-8 * abs(g["score"])


def test_closed_loop_ilqr_matches_reference():
    g = golden("loop_ilqr")
    nx = int(g["nx"])
    system = make_system(nx, 1, dt=float(g["dt"]))
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    model = MLPOracle(system, p)
    cost = cost_from_golden(g)
    ctl = ILQROracle(model, cost, float(g["dt"]), int(g["H"]))
    ctl.state_dim = nx + 1
    obs, ctrls = simulate(ctl, g["init"], model, 15)
    assert rel_err(obs, g["obs"]) < 1e-6 and rel_err(ctrls, g["ctrls"]) < 1e-6
    assert abs(cost.traj_cost(obs, ctrls) - g["score"]) < 1e-6 * abs(g["score"])


def _evalcfg_setup(g):
    nx = int(g["nx"])
    system = make_system(nx, 1, dt=0.05)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    return system, MLPOracle(system, p), cost_from_golden(g)


@pytest.mark.parametrize("name", ["loop_evalcfg_mppi", "loop_evalcfg_term"])
def test_evalcfg_mppi_matches_reference(name):
    """The tuner's objective through eval_cfg's call shape (pipeline_tuner.py:213-258): reset()
    after construction, task.term_cond (num_steps rows = num_steps - 1 controls, or the user's
    condition), max_steps = num_steps; surrogate branch, then the true-dynamics branch continuing
    the same global noise stream."""
    g = golden(name)
    system, model, cost = _evalcfg_setup(g)
    T = int(g["num_steps"])
    tc = None
    if "term_thresh" in g.files:
        min_len, thresh = int(g["term_min_len"]), float(g["term_thresh"])
        tc = lambda obs, ctrls: len(obs) >= min_len and abs(obs[-1][0]) < thresh   # noqa: E731
    np.random.seed(int(g["np_seed"]))

    def controller():
        return MPPIOracle(model, cost, np.array([g["bounds"]]), horizon=int(g["H"]),
                          num_path=int(g["N"]), sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    s, obs, ctrls = eval_cfg_episode(controller(), g["init"], model, T, cost.traj_cost, term_cond=tc)
    assert obs.shape == g["surr_obs"].shape
    if tc is None:
        assert len(obs) == T                 # num_steps rows, num_steps - 1 controls
    assert rel_err(obs, g["surr_obs"]) < 1e-8 and rel_err(ctrls, g["surr_ctrls"]) < 1e-8
    assert abs(s - g["surr_cost"]) < 1e-8 * abs(g["surr_cost"])
    s2, obs2, ctrls2 = eval_cfg_episode(controller(), g["init"], model, T, cost.traj_cost, term_cond=tc,
                                        dynamics=lambda x, u: model.pred(x, u))
    assert obs2.shape == g["truedyn_obs"].shape
    assert rel_err(obs2, g["truedyn_obs"]) < 1e-8 and rel_err(ctrls2, g["truedyn_ctrls"]) < 1e-8
    assert abs(s2 - g["truedyn_cost"]) < 1e-8 * abs(g["truedyn_cost"])


def test_evalcfg_ilqr_matches_reference():
    g = golden("loop_evalcfg_ilqr")
    system, model, cost = _evalcfg_setup(g)
    ctl = ILQROracle(model, cost, float(g["dt"]), int(g["H"]))
    ctl.state_dim = int(g["nx"]) + 1
    s, obs, ctrls = eval_cfg_episode(ctl, g["init"], model, int(g["num_steps"]), cost.traj_cost)
    assert len(obs) == int(g["num_steps"])
    assert rel_err(obs, g["surr_obs"]) < 1e-6 and rel_err(ctrls, g["surr_ctrls"]) < 1e-6
    assert abs(s - g["surr_cost"]) < 1e-6 * abs(g["surr_cost"])


# ---- sums of quadratic costs (sum_cost.py:49-54; gen_golden.gen_sumcost) --------------------------
from oracle.costs import SumCostOracle        # noqa: E402


@pytest.mark.parametrize("kind", ["gauss", "dense", "three", "samegoal"])
def test_sumcost_matches_reference(kind):
    """cost_sum.npz: the reference's SumCost of QuadCosts -- QuadCostFactory + GaussRegFactory's
    (goal of the second term = mean of the data), dense terms with different goals, three terms,
    and a same-goal sum -- through all its eval_* entry points and Cost.__call__."""
    g = golden("cost_sum")
    k = lambda name: g[kind + "_" + name]                    # noqa: E731
    c = SumCostOracle.from_arrays(k("Qs"), k("Rs"), k("Fs"), k("goals"))
    assert bool(k("is_quad")) == (kind == "samegoal")       # sum_cost.py:84-93
    obs, ctrl = k("obs"), k("ctrl")
    assert abs(c.eval_obs_cost(obs) - k("obs_cost")) < 1e-12 * max(1.0, abs(k("obs_cost")))
    assert abs(c.eval_ctrl_cost(ctrl) - k("ctrl_cost")) < 1e-12
    assert abs(c.eval_term_obs_cost(obs) - k("term_cost")) < 1e-12 * max(1.0, abs(k("term_cost")))
    for got, key in zip(c.eval_obs_cost_hess(obs), ("obs_c", "obs_j", "obs_h")):
        np.testing.assert_allclose(got, k(key), rtol=1e-13, atol=1e-12)
    for got, key in zip(c.eval_ctrl_cost_hess(ctrl), ("ctrl_c", "ctrl_j", "ctrl_h")):
        np.testing.assert_allclose(got, k(key), rtol=1e-13, atol=1e-12)
    for got, key in zip(c.eval_term_obs_cost_hess(obs), ("term_c", "term_j", "term_h")):
        np.testing.assert_allclose(got, k(key), rtol=1e-13, atol=1e-12)   # every term ignores its goal
    assert abs(c.traj_cost(k("traj_obs"), k("traj_ctrls")) - k("traj_cost")) < 1e-11 * abs(k("traj_cost"))


def test_evalcfg_sumcost_matches_reference():
    """eval_cfg's call shape with a QuadCostFactory + GaussRegFactory CONTROLLER cost (the pipeline
    hands the controller a task carrying the factory's cost, pipeline.py:156-160) while the episode
    is scored with the task's own cost: MPPI and iLQR."""
    g = golden("loop_evalcfg_sumcost")
    system, model, ctl_cost = _evalcfg_setup(g)
    task_cost = QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"])
    np.random.seed(int(g["np_seed"]))
    ctl = MPPIOracle(model, ctl_cost, np.array([g["bounds"]]), horizon=int(g["H"]), num_path=int(g["N"]),
                     sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    s, obs, ctrls = eval_cfg_episode(ctl, g["init"], model, int(g["num_steps"]), task_cost.traj_cost)
    assert rel_err(obs, g["surr_obs"]) < 1e-8 and rel_err(ctrls, g["surr_ctrls"]) < 1e-8
    assert abs(s - g["surr_cost"]) < 1e-8 * abs(g["surr_cost"])
    assert abs(ctl_cost.traj_cost(obs, ctrls) - g["ctl_cost_of_traj"]) < 1e-8 * abs(g["ctl_cost_of_traj"])
    g = golden("loop_evalcfg_sumcost_ilqr")
    system, model, ctl_cost = _evalcfg_setup(g)
    ctl = ILQROracle(model, ctl_cost, float(g["dt"]), int(g["H"]))
    ctl.state_dim = int(g["nx"]) + 1
    s, obs, ctrls = eval_cfg_episode(ctl, g["init"], model, int(g["num_steps"]), task_cost.traj_cost)
    assert rel_err(obs, g["surr_obs"]) < 1e-6 and rel_err(ctrls, g["surr_ctrls"]) < 1e-6
    assert abs(s - g["surr_cost"]) < 1e-6 * abs(g["surr_cost"])


@pytest.mark.parametrize("name", ["mlp_hc6_relu", "mlp_cp3_selu", "mlp_deep4_tanh", "mlp_odd1_sigmoid"])
def test_torch_structured_oracle_matches_reference(name):
    """MLPOracleTorch -- the reference's own call structure (torch f64 nn.Linear, per-column
    normalisation loops, autograd Jacobians), used by bench.py's CPU baseline -- against the same goldens."""
    from oracle.mlp import MLPOracleTorch
    g = golden(name)
    nx, nu = int(g["nx"]), int(g["nu"])
    p = golden_params(nx, nu, g["hidden"], g["activation"], g["seed"])
    m = MLPOracleTorch(make_system(nx, nu), p)
    assert rel_err(m.pred_batch(g["states"], g["ctrls"]), g["pred_batch"]) < 1e-11
    o, jx, ju = m.pred_diff_batch(g["states"], g["ctrls"])
    assert rel_err(o, g["diff_pred"]) < 1e-11 and rel_err(jx, g["diff_jx"]) < 1e-10 and rel_err(ju, g["diff_ju"]) < 1e-10
    assert rel_err(m.pred(g["states"][0], g["ctrls"][0]), g["pred0"]) < 1e-11
