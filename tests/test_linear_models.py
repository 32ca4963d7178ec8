"""Linear system-ID models (ARX, Koopman) -- SURVEY.md section 8 row f3.

CPU: the oracle restatement and the host classes' fitting / state construction against vectors
made by the reference's own ARX and Koopman (gen_golden.py gen_linear).  GPU: prediction,
Jacobians, MPPI and iLQR on the device model against the same vectors."""
import numpy as np
import pytest

from conftest import golden
from helpers import make_system, rel_err
from oracle.closed_loop import simulate as oracle_simulate
from oracle.costs import QuadCostOracle
from oracle.ilqr import ILQROracle
from oracle.linear import ARXOracle, KoopmanOracle
from oracle.mppi import MPPIOracle

CASES = ["arx3", "koop_full", "koop_lasso"]
CFG = {
    "arx3": dict(history=3),
    "koop_full": dict(method="lstsq", poly_basis="true", poly_degree=3, trig_basis="true",
                      trig_freq=2, product_terms="false"),
    "koop_lasso": dict(method="lasso", lasso_alpha=1e-4, poly_basis="true", poly_degree=2,
                       trig_basis="false", product_terms="false"),
}
FIT_TOL = {"arx3": 1e-8, "koop_full": 1e-6, "koop_lasso": 1e-6}


class _Traj:
    def __init__(self, obs, ctrls):
        self.obs, self.ctrls = obs, ctrls


def _oracle(tag, g, trained=True):
    system = make_system(3, 1, dt=float(g["dt"]))
    c = CFG[tag]
    if tag.startswith("arx"):
        m = ARXOracle(system, c["history"])
        if trained:
            m.train(list(g["train_obs"]), list(g["train_ctrls"]))
    else:
        m = KoopmanOracle(system, c["poly_basis"] == "true", c["poly_degree"], c["trig_basis"] == "true")
        if trained:
            m.train(list(g["train_obs"]), list(g["train_ctrls"]), c["method"], c.get("lasso_alpha"))
    return system, m


@pytest.mark.parametrize("tag", CASES)
def test_oracle_fit_and_state_logic_match_reference(tag):
    g = golden("linear_" + tag)
    system, m = _oracle(tag, g)
    assert m.state_dim == int(g["state_dim"])
    assert rel_err(m.A, g["A"]) < FIT_TOL[tag] and rel_err(m.B, g["B"]) < FIT_TOL[tag]
    m.A, m.B = g["A"], g["B"]          # everything below on the reference's own matrices
    obs0, ctl0 = g["train_obs"][0], g["train_ctrls"][0]
    assert rel_err(m.traj_to_state(_Traj(obs0[:7], ctl0[:7])), g["state_prefix7"]) < 1e-13
    assert rel_err(m.traj_to_state(_Traj(obs0[:1], ctl0[:1])), g["state_prefix1"]) < 1e-13
    np.testing.assert_allclose(m.state_from_first_obs(np.array([0.4, -0.3, 0.2]))[:3], [0.4, -0.3, 0.2])
    obs1, ctl1 = g["train_obs"][1], g["train_ctrls"][1]
    all_states = np.array([m.traj_to_state(_Traj(obs1[:t + 1], ctl1[:t + 1])) for t in range(len(obs1))])
    assert rel_err(all_states, g["states_traj1"]) < 1e-13
    got = m.update_state(g["pb_states"][3], g["pb_ctrls"][3], g["upd_in_obs"])
    assert rel_err(got, g["upd_state"]) < 1e-13
    assert rel_err(m.pred_batch(g["pb_states"], g["pb_ctrls"]), g["pred_batch"]) < 1e-13
    assert rel_err(m.pred(g["pb_states"][0], g["pb_ctrls"][0]), g["pred0"]) < 1e-13
    o, a, b = m.pred_diff(g["pb_states"][0], g["pb_ctrls"][0])
    assert rel_err(o, g["diff0_pred"]) < 1e-13
    np.testing.assert_array_equal(a, g["diff0_jx"])
    np.testing.assert_array_equal(b, g["diff0_ju"])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_solvers_on_linear_models_match_reference(tag):
    g = golden("linear_" + tag)
    system, m = _oracle(tag, g, trained=False)
    m.A, m.B = g["A"], g["B"]
    cost = QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"])
    np.random.seed(int(g["np_seed"]))
    ctl = MPPIOracle(m, cost, np.array([[-1.0, 1.0]]), horizon=int(g["H"]), num_path=int(g["N"]),
                     sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    np.testing.assert_allclose(ctl.act_sequence, g["mppi_act0"], rtol=0, atol=0)
    lift = m.state_from_first_obs
    obs, ctrls = oracle_simulate(ctl, g["init"], m, 6,
                                 traj_to_constate=lambda o: np.concatenate([lift(o), np.zeros(1)]))
    assert rel_err(obs, g["mppi_obs"]) < 1e-9 and rel_err(ctrls, g["mppi_ctrls"]) < 1e-9
    assert abs(cost.traj_cost(obs, ctrls) - g["mppi_score"]) < 1e-8 * abs(g["mppi_score"])
    il = ILQROracle(m, cost, float(g["dt"]), int(g["ilqr_H"]))
    conv, st, ct, Ks, ks = il.solve(g["ilqr_x0"], np.zeros((int(g["ilqr_H"]), 1)))
    assert conv == bool(g["ilqr_converged"])
    assert rel_err(st, g["ilqr_states"]) < 1e-7 and rel_err(ct, g["ilqr_ctrls"]) < 1e-7
    assert rel_err(Ks, g["ilqr_Ks"]) < 1e-6
    if "ilqr_loop_obs" in g:
        il2 = ILQROracle(m, cost, float(g["dt"]), int(g["ilqr_H"]))
        obs, ctrls = oracle_simulate(il2, g["init"], m, 5, traj_to_constate=lift)
        assert rel_err(obs, g["ilqr_loop_obs"]) < 1e-7 and rel_err(ctrls, g["ilqr_loop_ctrls"]) < 1e-6


def _host_model(tag, g, **kw):
    from autompc_amd import ARX, Koopman
    system = make_system(3, 1, dt=float(g["dt"]))
    cls = ARX if tag.startswith("arx") else Koopman
    return system, cls(system, **CFG[tag], **kw)


def _trajs(system, g):
    from autompc_amd import Trajectory
    return [Trajectory(system, o.shape[0], o.copy(), c.copy())
            for o, c in zip(g["train_obs"], g["train_ctrls"])]


@pytest.mark.parametrize("tag", CASES)
def test_host_classes_fit_and_state_logic_match_reference(tag):
    g = golden("linear_" + tag)
    system, m = _host_model(tag, g)
    trajs = _trajs(system, g)
    m.train(trajs)
    assert m.state_dim == int(g["state_dim"]) and m.is_linear and m.is_diff
    assert rel_err(m.A, g["A"]) < FIT_TOL[tag] and rel_err(m.B, g["B"]) < FIT_TOL[tag]
    p = m.get_parameters()
    m2 = _host_model(tag, g)[1]
    m2.set_parameters(p)
    np.testing.assert_array_equal(m2.A, m.A)
    np.testing.assert_array_equal(m2.B, m.B)
    if tag.startswith("arx"):
        m.set_parameters({"coeffs": np.concatenate([g["A"][:3], g["B"][:3]], axis=1)})
        np.testing.assert_array_equal(m.A, g["A"])
        np.testing.assert_array_equal(m.B, g["B"])
    else:
        m.set_parameters({"A": g["A"], "B": g["B"]})
    assert rel_err(m.traj_to_state(trajs[0][:7]), g["state_prefix7"]) < 1e-13
    assert rel_err(m.traj_to_state(trajs[0][:1]), g["state_prefix1"]) < 1e-13
    assert rel_err(m.traj_to_states(trajs[1]), g["states_traj1"]) < 1e-13
    got = m.update_state(g["pb_states"][3], g["pb_ctrls"][3], g["upd_in_obs"])
    assert rel_err(got, g["upd_state"]) < 1e-13
    a, b = m.to_linear()
    np.testing.assert_array_equal(a, g["A"])
    a[0, 0] += 1.0
    assert m.A[0, 0] == g["A"][0, 0]                       # copies, as the reference returns


def test_koopman_documented_basis_and_validation():
    from autompc_amd import Koopman
    system = make_system(2, 1)
    strict = Koopman(system, poly_basis=True, poly_degree=3, trig_basis=True, trig_freq=2)
    assert strict.basis == [(0, 1), (1, 3), (1, 3), (2, 3), (3, 3), (2, 3), (3, 3), (2, 3), (3, 3)]
    fixed = Koopman(system, poly_basis=True, poly_degree=3, trig_basis=True, trig_freq=2,
                    strict_reference=False)
    assert fixed.basis == [(0, 1), (1, 2), (1, 3), (2, 1), (3, 1), (2, 2), (3, 2)]
    x = np.array([0.3, -0.5])
    np.testing.assert_allclose(fixed._apply_basis(x)[2:6], np.concatenate([x ** 2, x ** 3]))
    with pytest.raises(ValueError):
        Koopman(system, method="bogus")
    with pytest.raises(NotImplementedError):
        Koopman(system, method="stable").train([])  # noqa


# ------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("tag", CASES)
def test_device_prediction_and_jacobians(tag, precision):
    g = golden("linear_" + tag)
    _, m = _host_model(tag, g, precision=precision)
    if tag.startswith("arx"):
        m.set_parameters({"coeffs": np.concatenate([g["A"][:3], g["B"][:3]], axis=1)})
    else:
        m.set_parameters({"A": g["A"], "B": g["B"]})
    tol = 1e-13 if precision == "f64" else 3e-6
    assert rel_err(m.pred_batch(g["pb_states"], g["pb_ctrls"]), g["pred_batch"]) < tol
    assert rel_err(m.pred(g["pb_states"][0], g["pb_ctrls"][0]), g["pred0"]) < tol
    o, jx, ju = m.pred_diff_batch(g["pb_states"], g["pb_ctrls"])
    assert rel_err(o, g["pred_batch"]) < tol
    for i in range(jx.shape[0]):
        assert rel_err(jx[i], g["A"]) < tol and rel_err(ju[i], g["B"]) < tol
    o0, a0, b0 = m.pred_diff(g["pb_states"][0], g["pb_ctrls"][0])
    assert rel_err(o0, g["diff0_pred"]) < tol and rel_err(a0, g["diff0_jx"]) < tol


def _task(system, g, bounded):
    from autompc_amd import QuadCost, Task
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    if bounded:
        task.set_ctrl_bound("u0", -1.0, 1.0)
    return task


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_device_mppi_closed_loop_matches_reference(tag):
    from autompc_amd import MPPI, simulate
    g = golden("linear_" + tag)
    system, m = _host_model(tag, g)
    m.set_parameters({"coeffs": np.concatenate([g["A"][:3], g["B"][:3]], axis=1)}
                     if tag.startswith("arx") else {"A": g["A"], "B": g["B"]})
    task = _task(system, g, True)
    np.random.seed(int(g["np_seed"]))
    ctl = MPPI(system, task, m, horizon=int(g["H"]), num_path=int(g["N"]), sigma=float(g["sigma"]),
               lmda=float(g["lmda"]))
    np.testing.assert_allclose(ctl.act_sequence, g["mppi_act0"], rtol=0, atol=0)
    traj = simulate(ctl, g["init"], sim_model=m, max_steps=6)
    assert rel_err(traj.obs, g["mppi_obs"]) < 1e-8 and rel_err(traj.ctrls, g["mppi_ctrls"]) < 1e-8
    assert abs(task.get_cost()(traj) - g["mppi_score"]) < 1e-7 * abs(g["mppi_score"])


@pytest.mark.gpu
def test_candidate_evaluator_on_arx_matches_reference_simulate():
    """The reference's simulate() with MPPI on its fitted ARX model (history 3: the model state is
    the stacked history, not the observation) replayed through the device-resident evaluator; and
    on its Koopman models, whose controller state is re-lifted from every observation."""
    from autompc_amd.tuning import CandidateEvaluator
    g = golden("linear_arx3")
    system, m = _host_model("arx3", g)
    m.set_parameters({"coeffs": np.concatenate([g["A"][:3], g["B"][:3]], axis=1)})
    task = _task(system, g, True)
    N, H, T, scale = int(g["N"]), int(g["H"]), 6, np.sqrt(float(g["sigma"]))
    np.random.seed(int(g["np_seed"]))
    act0 = np.random.normal(scale=scale, size=(H, 1))
    eps = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(T)])
    ev = CandidateEvaluator(system, task, m)
    cand = dict(horizon=H, sigma=float(g["sigma"]), lmda=float(g["lmda"]), num_path=N, Q=g["Q"], R=g["R"], F=g["F"])
    scores, obs, ctrls = ev.evaluate([cand], n_steps=T, init_obs=g["init"], eps_all=eps, act_init=act0,
                                     return_trajectories=True)
    assert obs.shape[2] == m.state_dim
    assert rel_err(obs[0][:, :3], g["mppi_obs"]) < 1e-8 and rel_err(ctrls[0], g["mppi_ctrls"]) < 1e-8
    assert abs(scores[0] - g["mppi_score"]) < 1e-7 * abs(g["mppi_score"])
    # Koopman re-lifts its state from every observation (koopman.py:166-168): the device loop's state
    # lift (round 4; refused before) against the reference's simulate() on both fitted Koopman models
    for tag in ("koop_full", "koop_lasso"):
        gk = golden("linear_" + tag)
        systemk, mk = _host_model(tag, gk)
        mk.set_parameters({"A": gk["A"], "B": gk["B"]})
        np.random.seed(int(gk["np_seed"]))
        act0 = np.random.normal(scale=scale, size=(H, 1))
        eps = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(T)])
        evk = CandidateEvaluator(systemk, _task(systemk, gk, True), mk)
        cand = dict(horizon=H, sigma=float(gk["sigma"]), lmda=float(gk["lmda"]), num_path=N, Q=gk["Q"], R=gk["R"], F=gk["F"])
        scores, obs, ctrls = evk.evaluate([cand], n_steps=T, init_obs=gk["init"], eps_all=eps, act_init=act0,
                                          return_trajectories=True)
        assert obs.shape[2] == mk.state_dim
        assert rel_err(obs[0][:, :3], gk["mppi_obs"]) < 1e-8 and rel_err(ctrls[0], gk["mppi_ctrls"]) < 1e-8
        assert abs(scores[0] - gk["mppi_score"]) < 1e-7 * abs(gk["mppi_score"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_device_ilqr_matches_reference(tag):
    from autompc_amd import IterativeLQR, simulate
    g = golden("linear_" + tag)
    system, m = _host_model(tag, g)
    m.set_parameters({"coeffs": np.concatenate([g["A"][:3], g["B"][:3]], axis=1)}
                     if tag.startswith("arx") else {"A": g["A"], "B": g["B"]})
    H = int(g["ilqr_H"])
    ctl = IterativeLQR(system, _task(system, g, False), m, H)
    conv, st, ct, Ks, ks = ctl.compute_ilqr_default(g["ilqr_x0"], np.zeros((H, 1)))
    assert conv == bool(g["ilqr_converged"])
    assert rel_err(st, g["ilqr_states"]) < 1e-6 and rel_err(ct, g["ilqr_ctrls"]) < 1e-6
    assert rel_err(Ks, g["ilqr_Ks"]) < 1e-5
    # closed loop; on ARX this is a path the reference itself cannot run (see IterativeLQR
    # .traj_to_state), so only the Koopman cases have reference vectors
    traj = simulate(ctl, g["init"], sim_model=m, max_steps=5)
    if "ilqr_loop_obs" in g:
        assert rel_err(traj.obs, g["ilqr_loop_obs"]) < 1e-6
        assert rel_err(traj.ctrls, g["ilqr_loop_ctrls"]) < 1e-5
    else:
        assert traj.obs.shape == (6, 3) and np.all(np.isfinite(traj.obs))


@pytest.mark.gpu
@pytest.mark.parametrize("ns,nu,no", [(41, 6, 17), (64, 16, 5), (33, 1, 33)])
def test_wide_linear_states(ns, nu, no):
    """33..64 model states (ARX history 2 on a 17-observation / 6-control system is 41): the four-output-tile
    MFMA path; prediction, Jacobians and an MPPI solve against the oracle."""
    from autompc_amd import MPPI, QuadCost, Task, _lib
    from oracle.linear import LinearOracle
    rng = np.random.default_rng(ns)
    S = rng.normal(size=(ns, ns))
    A = 0.9 * np.eye(ns) + 0.1 * (S - S.T) / np.sqrt(ns)
    B = rng.normal(scale=0.3, size=(ns, nu))
    h = _lib.Handle(0, "f64")
    h.set_linear(A, B)
    s_, c_ = rng.normal(size=(70, ns)), rng.normal(size=(70, nu))
    assert rel_err(h.pred_batch(s_, c_), s_ @ A.T + c_ @ B.T) < 1e-13
    o, jx, ju = h.pred_diff_batch(s_[:5], c_[:5])
    assert rel_err(o, s_[:5] @ A.T + c_[:5] @ B.T) < 1e-13
    for k in range(5):
        np.testing.assert_array_equal(jx[k], A)
        np.testing.assert_array_equal(ju[k], B)
    h.set_quad_costs(np.eye(no), np.eye(nu), np.eye(no), np.zeros(no))
    if ns + nu + 1 > 64:                                # one wave holds the augmented Quu system
        with pytest.raises(_lib.AmpcError):
            _lib.IlqrPlan(h, 1, 5, 0.05)
    h.close()
    # MPPI through the plugin classes
    system = make_system(no, nu)

    class Carrier:                       # minimal device-stageable model around (A, B)
        def __init__(self):
            self.system, self.state_dim, self.precision, self.device = system, ns, "f64", 0

        def stage_into(self, handle):
            handle.set_linear(A, B)

        def update_state(self, state, ctrl, obs):
            return np.concatenate([np.asarray(obs), np.asarray(state)[no:]])

        def traj_to_state(self, traj):
            return np.concatenate([traj[-1].obs, np.zeros(ns - no)])
    orc_m = LinearOracle(system, A, B)
    orc_m.state_dim = ns
    orc_m.update_state = Carrier().update_state
    Q, R, F = np.diag(rng.uniform(0.5, 2, size=no)), np.diag(rng.uniform(0.01, 0.1, size=nu)), np.eye(no)
    goal = rng.normal(scale=0.1, size=no)
    task = Task(system)
    task.set_cost(QuadCost(system, Q, R, F, goal=goal))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    np.random.seed(3)
    orc = MPPIOracle(orc_m, QuadCostOracle(Q, R, F, goal), np.tile([-1.0, 1.0], (nu, 1)), horizon=9,
                     num_path=150, sigma=0.5, lmda=0.8)
    np.random.seed(3)
    ctl = MPPI(system, task, Carrier(), horizon=9, num_path=150, sigma=0.5, lmda=0.8)
    obs = rng.uniform(-0.5, 0.5, size=no)
    cs = np.concatenate([obs, rng.uniform(-0.2, 0.2, size=ns - no), np.zeros(nu)])
    st = np.random.get_state()
    uo, cso = orc.run(cs, obs)
    np.random.set_state(st)
    uh, csh = ctl.run(cs, obs, return_details=True)
    assert rel_err(ctl.last_costs, orc.last_costs) < 1e-10
    assert rel_err(uh, uo) < 1e-9 and rel_err(csh, cso) < 1e-9


@pytest.mark.gpu
def test_device_rejects_oversized_linear_state():
    from autompc_amd import _lib
    h = _lib.Handle(0, "f64")
    with pytest.raises(_lib.AmpcError):
        h.set_linear(np.eye(257), np.zeros((257, 1)))        # (65..256: csrc/linear_kernels.hpp, test_linear_wide.py)
    h.set_linear(np.eye(65), np.zeros((65, 1)))
    h.close()


@pytest.mark.gpu
def test_arx_history_two_on_a_halfcheetah_sized_system_closed_loop():
    """The case the 32-state limit used to exclude: 17 observations, 6 controls, history 2 ->
    41 model states.  Host ARX class (fit, state bookkeeping) + device MPPI, against the oracle's
    ARX + MPPI in lock-step over a short closed loop."""
    from autompc_amd import ARX, MPPI, QuadCost, Task, Trajectory
    no, nu, k = 17, 6, 2
    system = make_system(no, nu)
    rng = np.random.default_rng(12)
    M = 0.92 * np.eye(no) + 0.05 * rng.normal(size=(no, no)) / np.sqrt(no)
    G = rng.normal(scale=0.2, size=(no, nu))
    trajs = []
    for _ in range(6):
        obs = np.zeros((60, no))
        ctl = rng.uniform(-1, 1, size=(60, nu))
        x = rng.uniform(-1, 1, size=no)
        for t in range(60):
            obs[t] = x
            x = M @ x + G @ ctl[t] + 0.02 * np.sin(x[::-1])
        trajs.append(Trajectory(system, 60, obs, ctl))
    model = ARX(system, history=k)
    model.train(trajs)
    assert model.state_dim == 41
    oracle_m = ARXOracle(system, k, model.A, model.B)
    task = Task(system)
    Q, R, F = np.eye(no), 0.05 * np.eye(nu), 2.0 * np.eye(no)
    task.set_cost(QuadCost(system, Q, R, F))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    np.random.seed(5)
    orc = MPPIOracle(oracle_m, QuadCostOracle(Q, R, F, np.zeros(no)), np.tile([-1.0, 1.0], (nu, 1)),
                     horizon=10, num_path=200, sigma=0.6, lmda=0.7)
    np.random.seed(5)
    ctl = MPPI(system, task, model, horizon=10, num_path=200, sigma=0.6, lmda=0.7)
    obs = rng.uniform(-0.5, 0.5, size=no)
    one = Trajectory(system, 1, obs[None, :].copy(), np.zeros((1, nu)))
    cs_h = ctl.traj_to_state(one)
    cs_o = np.concatenate([oracle_m.state_from_first_obs(obs), np.zeros(nu)])
    np.testing.assert_allclose(cs_h, cs_o, rtol=0, atol=0)
    for _ in range(4):
        st = np.random.get_state()
        uo, cs_o = orc.run(cs_o, obs)
        np.random.set_state(st)
        uh, cs_h = ctl.run(cs_h, obs)
        assert rel_err(uh, uo) < 1e-8 and rel_err(cs_h, cs_o) < 1e-8
        ctl.act_sequence = orc.act_sequence          # lock-step: per-solve error only
        cs_h = cs_o.copy()
        obs = M @ obs + G @ uo


# ------------------------------------------------------------ model states above 32 (MFMA path)
def _wide_models(precision="f64"):
    """ARX with history 2 on a HalfCheetah-sized system: 41 model states (linear_arx2_wide.npz)."""
    from autompc_amd import ARX
    g = golden("linear_arx2_wide")
    system = make_system(17, 6, dt=float(g["dt"]))
    m = ARX(system, history=2, precision=precision)
    m.set_parameters({"coeffs": np.concatenate([g["A"][:17], g["B"][:17]], axis=1)})
    orc = ARXOracle(system, 2, g["A"], g["B"])
    return g, system, m, orc


def test_wide_oracle_fit_matches_reference():
    g = golden("linear_arx2_wide")
    system = make_system(17, 6, dt=float(g["dt"]))
    m = ARXOracle(system, 2)
    m.train(list(g["train_obs"]), list(g["train_ctrls"]))
    assert m.state_dim == 41
    assert rel_err(m.A, g["A"]) < 1e-7 and rel_err(m.B, g["B"]) < 1e-7
    m.A, m.B = g["A"], g["B"]
    assert rel_err(m.pred_batch(g["pb_states"], g["pb_ctrls"]), g["pred_batch"]) < 1e-13
    orc = ILQROracle(m, QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"]), float(g["dt"]), int(g["ilqr_H"]))
    conv, st, ct, Ks, ks = orc.solve(g["ilqr_x0"], np.zeros((int(g["ilqr_H"]), 6)))
    assert conv == bool(g["ilqr_converged"])
    assert rel_err(st, g["ilqr_states"]) < 1e-6 and rel_err(ct, g["ilqr_ctrls"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_wide_device_prediction_and_jacobians(precision):
    g, system, m, _ = _wide_models(precision)
    assert m.A.shape == (41, 41)
    tol = 1e-13 if precision == "f64" else 3e-6
    assert rel_err(m.pred_batch(g["pb_states"], g["pb_ctrls"]), g["pred_batch"]) < tol
    o, jx, ju = m.pred_diff_batch(g["pb_states"], g["pb_ctrls"])
    assert rel_err(o, g["pred_batch"]) < tol
    for i in range(jx.shape[0]):
        assert rel_err(jx[i], g["A"]) < tol and rel_err(ju[i], g["B"]) < tol
    o0, a0, b0 = m.pred_diff(g["pb_states"][0], g["pb_ctrls"][0])
    assert rel_err(o0, g["diff0_pred"]) < tol and rel_err(a0, g["diff0_jx"]) < tol and rel_err(b0, g["diff0_ju"]) < tol


@pytest.mark.gpu
def test_wide_device_ilqr_matches_reference():
    from autompc_amd import IterativeLQR
    g, system, m, _ = _wide_models()
    H = int(g["ilqr_H"])
    ctl = IterativeLQR(system, _task(system, g, False), m, H)
    conv, st, ct, Ks, ks = ctl.compute_ilqr_default(g["ilqr_x0"], np.zeros((H, 6)))
    assert conv == bool(g["ilqr_converged"])
    assert rel_err(st, g["ilqr_states"]) < 1e-6 and rel_err(ct, g["ilqr_ctrls"]) < 1e-6
    assert rel_err(Ks, g["ilqr_Ks"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tile_rows", [16, 32, 64])
def test_wide_device_mppi_matches_oracle(tile_rows, monkeypatch):
    """MPPI with 6 controls on the 41-state model (the reference's MPPI cannot run nu > 1) vs the
    oracle, for every tile height of the wide MFMA path."""
    from autompc_amd import MPPI
    monkeypatch.setenv("AMPC_MT", str(tile_rows // 16))
    g, system, m, orc_model = _wide_models()
    from autompc_amd import QuadCost, Task
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    task.set_ctrl_bounds(-np.ones(6), np.ones(6))
    N, H = 200, 9
    np.random.seed(3)
    ctl = MPPI(system, task, m, horizon=H, num_path=N, sigma=0.5, lmda=0.8)
    np.random.seed(3)
    orc = MPPIOracle(orc_model, QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"]), np.tile([-1.0, 1.0], (6, 1)),
                     horizon=H, num_path=N, sigma=0.5, lmda=0.8)
    x = g["ilqr_x0"]
    cs_h = cs_o = np.concatenate([x, np.zeros(6)])
    obs = g["init"]
    for _ in range(2):
        st = np.random.get_state()
        uo, cs_o = orc.run(cs_o, obs)
        np.random.set_state(st)
        uh, cs_h = ctl.run(cs_h, obs, return_details=True)
        assert rel_err(ctl.last_costs, orc.last_costs) < 1e-9
        assert rel_err(ctl.act_sequence, orc.act_sequence) < 1e-8 and rel_err(uh, uo) < 1e-8
        ctl.act_sequence = orc.act_sequence
        obs = orc_model.pred(cs_o[:41], uo)[:17]
