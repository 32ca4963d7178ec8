"""Linear models beyond 64 states and lifted models in the device closed loop (VERDICT r3 missing 3).

ARX keeps `history` (1..10, arx.py:27,37-45) observations and controls in its state: on a
HalfCheetah-sized system (18 observations, 6 controls) history 10 is 235 states -- the dedicated
K-tiled MFMA kernels (csrc/linear_kernels.hpp).  Koopman rebuilds its state from every observation
(koopman.py:166-168) -- the device closed loop's state lift (ampc_mppi_plan_set_state_lift).
Fixtures: the reference's own ARX / Koopman / MPPI / simulate (gen_golden.py gen_linear_wide2,
gen_evalcfg_koopman).  CPU: oracle + host classes; GPU: the device paths."""
import numpy as np
import pytest

from conftest import golden
from helpers import check_weights, golden_params, make_system, rel_err
from oracle.closed_loop import eval_cfg_episode, simulate as oracle_simulate
from oracle.costs import QuadCostOracle
from oracle.linear import ARXOracle, KoopmanOracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle

WIDE = {"arx10_hc": 6, "arx10_nu1": 1}


class _Traj:
    def __init__(self, obs, ctrls):
        self.obs, self.ctrls = obs, ctrls


def _full(m, g):
    """The reference's A, B from the stored dense rows (the rest is the history shift, arx.py:121-148)
    -- checked against the stored probes / sums of the reference's own matrices."""
    no = 18
    k = g["coeffs"].shape[1]
    ns = int(g["state_dim"])
    A, B = m._build_system_matrices(g["coeffs"]) if hasattr(m, "_build_system_matrices") else m.build(g["coeffs"])
    assert A.shape == (ns, ns) and k == ns + B.shape[1]
    idx = g["A_probe_idx"]
    np.testing.assert_array_equal(A[idx[:, 0], idx[:, 1]], g["A_probe"])
    assert abs(A.sum() - g["A_sum"]) < 1e-9 * g["A_abs_sum"] and abs(np.abs(B).sum() - g["B_abs_sum"]) < 1e-12 * g["B_abs_sum"]
    np.testing.assert_array_equal(A[:no], g["coeffs"][:, :ns])
    return A, B


@pytest.mark.parametrize("tag", list(WIDE))
def test_host_arx_history_10_matches_reference(tag):
    from autompc_amd import ARX, Trajectory
    g = golden("linear_" + tag)
    nu = WIDE[tag]
    system = make_system(18, nu)
    m = ARX(system, history=10)
    assert m.state_dim == int(g["state_dim"]) == 10 * 18 + 9 * nu + 1
    trajs = [Trajectory(system, o.shape[0], o.copy(), c.copy()) for o, c in zip(g["train_obs"], g["train_ctrls"])]
    m.train(trajs)
    assert rel_err(np.concatenate([m.A[:18], m.B[:18]], axis=1), g["coeffs"]) < 1e-6     # (min-norm lstsq, 235 unknowns)
    m.set_parameters({"coeffs": g["coeffs"]})
    _full(m, g)
    assert rel_err(m.traj_to_state(trajs[0][:12]), g["state_prefix12"]) < 1e-13
    assert rel_err(m.traj_to_state(trajs[0][:1]), g["state_prefix1"]) < 1e-13
    # the oracle restatement, on the same matrices
    o = ARXOracle(system, 10)
    o.A, o.B = m.A.copy(), m.B.copy()
    assert rel_err(o.pred_batch(g["pb_states"], g["pb_ctrls"]), g["pred_batch"]) < 1e-12
    assert rel_err(o.traj_to_state(_Traj(g["train_obs"][0][:12], g["train_ctrls"][0][:12])), g["state_prefix12"]) < 1e-13


def test_oracle_mppi_on_the_190_state_arx_matches_reference():
    g = golden("linear_arx10_nu1")
    system = make_system(18, 1)
    from autompc_amd import ARX
    h = ARX(system, history=10)
    h.set_parameters({"coeffs": g["coeffs"]})
    m = ARXOracle(system, 10)
    m.A, m.B = h.A.copy(), h.B.copy()
    cost = QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"])
    np.random.seed(int(g["np_seed"]))
    ctl = MPPIOracle(m, cost, np.array([[-1.0, 1.0]]), horizon=int(g["H"]), num_path=int(g["N"]),
                     sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    np.testing.assert_array_equal(ctl.act_sequence, g["mppi_act0"])
    lift = m.state_from_first_obs
    obs, ctrls = oracle_simulate(ctl, g["init"], m, 6, traj_to_constate=lambda o: np.concatenate([lift(o), np.zeros(1)]))
    assert rel_err(obs, g["mppi_obs"]) < 1e-9 and rel_err(ctrls, g["mppi_ctrls"]) < 1e-9
    assert abs(cost.traj_cost(obs, ctrls) - g["mppi_score"]) < 1e-8 * abs(g["mppi_score"])


def _koop_oracle(g):
    system = make_system(3, 1)
    m = KoopmanOracle(system, True, 3, True)
    m.A, m.B = g["A"], g["B"]
    assert m.state_dim == int(g["state_dim"])
    return system, m


def test_oracle_koopman_evalcfg_matches_reference():
    """The controller re-lifts every observation while the simulation state advances linearly: both
    episodes of loop_evalcfg_koopman.npz (simulated on the Koopman model itself / on an MLP)."""
    g = golden("loop_evalcfg_koopman")
    system, m = _koop_oracle(g)
    p = golden_params(3, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    cost = QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"])
    for tag, sim in (("self", m), ("mlp", MLPOracle(system, p))):
        np.random.seed(int(g[tag + "_np_seed"]))
        ctl = MPPIOracle(m, cost, np.array([g["bounds"]]), horizon=int(g["H"]), num_path=int(g["N"]),
                         sigma=float(g["sigma"]), lmda=float(g["lmda"]))
        s, obs, ctrls = eval_cfg_episode(ctl, g["init"], sim, int(g["num_steps"]), cost.traj_cost,
                                         traj_to_constate=lambda o: np.concatenate([m.state_from_first_obs(o), np.zeros(1)]))
        assert rel_err(obs, g[tag + "_obs"]) < 1e-8 and rel_err(ctrls, g[tag + "_ctrls"]) < 1e-8
        assert abs(s - g[tag + "_cost"]) < 1e-8 * abs(g[tag + "_cost"])


# ------------------------------------------------------------------------------------------ GPU
def _dev_arx(tag, g, precision="f64"):
    from autompc_amd import ARX
    system = make_system(18, WIDE[tag])
    m = ARX(system, history=10, precision=precision)
    m.set_parameters({"coeffs": g["coeffs"]})
    return system, m


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("tag", list(WIDE))
def test_device_prediction_and_jacobians_at_235_states(tag, precision):
    g = golden("linear_" + tag)
    system, m = _dev_arx(tag, g, precision)
    tol = 1e-13 if precision == "f64" else 3e-6
    assert rel_err(m.pred_batch(g["pb_states"], g["pb_ctrls"]), g["pred_batch"]) < tol
    assert rel_err(m.pred(g["pb_states"][0], g["pb_ctrls"][0]), g["pred0"]) < tol
    o, jx, ju = m.pred_diff_batch(g["pb_states"][:5], g["pb_ctrls"][:5])
    assert rel_err(o, g["pred_batch"][:5]) < tol
    for i in range(5):
        assert rel_err(jx[i], m.A) < (0 if precision == "f64" else 1e-7) + 1e-300
        assert rel_err(ju[i], m.B) < (0 if precision == "f64" else 1e-7) + 1e-300
    # ragged batch (not a multiple of the 16-row tile)
    one = m.pred_batch(g["pb_states"][:19], g["pb_ctrls"][:19])
    assert rel_err(one, g["pred_batch"][:19]) < tol


@pytest.mark.gpu
def test_device_mppi_on_the_190_state_arx_matches_reference_simulate():
    """The reference's simulate() + MPPI on its history-10 ARX fit, through the drop-in controller
    (default noise mode) and through the device-resident candidate evaluator."""
    from autompc_amd import MPPI, QuadCost, Task, simulate
    from autompc_amd.tuning import CandidateEvaluator
    g = golden("linear_arx10_nu1")
    system, m = _dev_arx("arx10_nu1", g)
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    task.set_ctrl_bound("u0", -1.0, 1.0)
    N, H, T = int(g["N"]), int(g["H"]), 6
    np.random.seed(int(g["np_seed"]))
    ctl = MPPI(system, task, m, horizon=H, num_path=N, sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    np.testing.assert_array_equal(ctl.act_sequence, g["mppi_act0"])
    traj = simulate(ctl, g["init"], sim_model=m, max_steps=T)
    assert rel_err(traj.obs, g["mppi_obs"]) < 1e-8 and rel_err(traj.ctrls, g["mppi_ctrls"]) < 1e-8
    assert abs(task.get_cost()(traj) - g["mppi_score"]) < 1e-7 * abs(g["mppi_score"])
    scale = np.sqrt(float(g["sigma"]))
    np.random.seed(int(g["np_seed"]))
    act0 = np.random.normal(scale=scale, size=(H, 1))
    eps = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(T)])
    ev = CandidateEvaluator(system, task, m, tile_rows=16)
    cand = dict(horizon=H, sigma=float(g["sigma"]), lmda=float(g["lmda"]), num_path=N, Q=g["Q"], R=g["R"], F=g["F"])
    scores, obs, ctrls = ev.evaluate([cand], n_steps=T, init_obs=g["init"], eps_all=eps, act_init=act0,
                                     return_trajectories=True)
    assert obs.shape[2] == 190
    assert rel_err(obs[0][:, :18], g["mppi_obs"]) < 1e-8 and rel_err(ctrls[0], g["mppi_ctrls"]) < 1e-8
    assert abs(scores[0] - g["mppi_score"]) < 1e-7 * abs(g["mppi_score"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # ns, nu, no, N, H, precision, tol, dense cost, per-particle terminal
    (235, 6, 18, 300, 12, "f64", 1e-9, False, False),      # ARX history 10 on HalfCheetah
    (66, 6, 18, 77, 9, "f64", 1e-9, True, True),           # history 3: just past the tile's 64
    (256, 16, 32, 40, 5, "f64", 1e-9, True, False),        # the largest shape
    (129, 2, 5, 200, 10, "f32", 2e-4, False, False),
])
def test_wide_linear_mppi_vs_oracle(case):
    from autompc_amd import MPPI, QuadCost, Task
    from autompc_amd.sysid.model import Model
    ns, nu, no, N, H, precision, tol, dense, ppt = case
    rng = np.random.default_rng(ns)
    A = 0.9 * np.linalg.qr(rng.normal(size=(ns, ns)))[0] + 0.02 * rng.normal(size=(ns, ns)) / np.sqrt(ns)
    Bm = rng.normal(scale=0.3, size=(ns, nu))
    system = make_system(no, nu)

    class Lin(Model):
        precision_ = precision

        def __init__(self):
            super().__init__(system)
            self.precision, self.device, self._h = precision, 0, None

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
    if dense:
        W = rng.normal(size=(no, no))
        Q, F = W @ W.T / no + 0.1 * rng.normal(size=(no, no)), np.eye(no)
        R = np.diag(rng.uniform(0.01, 0.1, size=nu)) + 0.002
    else:
        Q, F, R = np.diag(rng.uniform(0.5, 2, size=no)), np.diag(rng.uniform(0.5, 2, size=no)), np.diag(rng.uniform(0.01, 0.1, size=nu))
    goal = rng.normal(scale=0.1, size=no)
    task = Task(system)
    task.set_cost(QuadCost(system, Q, R, F, goal=goal))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    model = Lin()
    np.random.seed(3)
    orc = MPPIOracle(model, QuadCostOracle(Q, R, F, goal), np.tile([-1.0, 1.0], (nu, 1)), horizon=H, num_path=N,
                     sigma=0.7, lmda=0.9, per_particle_terminal=ppt)
    np.random.seed(3)
    ctl = MPPI(system, task, model, horizon=H, num_path=N, sigma=0.7, lmda=0.9, per_particle_terminal=ppt,
               precision=precision)
    x = rng.uniform(-0.3, 0.3, size=ns)
    cs = np.concatenate([x, np.zeros(nu)])
    for _ in range(2):
        st = np.random.get_state()
        uo, cs_o = orc.run(cs, x[:no])
        np.random.set_state(st)
        uh, cs_h = ctl.run(cs, x[:no], return_details=True)
        assert rel_err(ctl.last_costs, orc.last_costs) < tol
        assert rel_err(ctl.act_sequence, orc.act_sequence) < tol * 10 and rel_err(uh, uo) < tol * 10
        ctl.act_sequence = orc.act_sequence


@pytest.mark.gpu
def test_ilqr_limits_on_wide_linear_models_are_stated():
    """iLQR on wide linear models: up to 128 states (ilqr_wide.hpp); beyond that a clear refusal."""
    from autompc_amd import _lib
    h = _lib.Handle(0, "f64")
    h.set_linear(0.5 * np.eye(140), np.ones((140, 2)))
    h.set_quad_costs(np.eye(5), np.eye(2), np.eye(5), np.zeros(5))
    with pytest.raises(_lib.AmpcError, match="up to 128"):
        _lib.IlqrPlan(h, 1, 5, 0.05)
    h.set_linear(0.5 * np.eye(70), np.ones((70, 5)))
    h.set_quad_costs(np.eye(5), np.eye(5), np.eye(5), np.zeros(5))
    with pytest.raises(_lib.AmpcError, match="1, 2, 3, 4, 6 or 8 controls"):
        _lib.IlqrPlan(h, 1, 5, 0.05)
    h.close()


def _koop_stack(g):
    from autompc_amd import Koopman, QuadCost, Task
    system = make_system(3, 1)
    m = Koopman(system, method="lstsq", poly_basis="true", poly_degree=3, trig_basis="true", trig_freq=2,
                product_terms="false")
    m.set_parameters({"A": g["A"], "B": g["B"]})
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    task.set_ctrl_bounds([g["bounds"][0]], [g["bounds"][1]])
    task.set_init_obs(g["init"])
    task.set_num_steps(int(g["num_steps"]))
    return system, m, task


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["self", "mlp"])
def test_candidate_evaluator_carries_a_lifted_controller_state(tag):
    """loop_evalcfg_koopman.npz: eval_cfg's episode with the reference's MPPI on its Koopman model,
    simulated on the model itself (21-dimensional simulation state advancing linearly, the controller
    re-lifting its first three entries every step) and on an MLP surrogate -- through the device
    evaluator (state lift) and through host simulate() + the drop-in controller."""
    from autompc_amd import MLP, MPPI, simulate
    from autompc_amd.tuning import CandidateEvaluator
    g = golden("loop_evalcfg_koopman")
    system, m, task = _koop_stack(g)
    sur = m
    if tag == "mlp":
        p = golden_params(3, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
        check_weights(p, g)
        sur = MLP(system, n_hidden_layers=2, hidden_size_1=48, hidden_size_2=48, nonlintype="tanh")
        sur.weights, sur.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
        sur.xu_means, sur.xu_std, sur.dy_means, sur.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    N, H, T = int(g["N"]), int(g["H"]), int(g["num_steps"])
    scale = np.sqrt(float(g["sigma"]))
    # host loop, default noise mode
    np.random.seed(int(g[tag + "_np_seed"]))
    ctl = MPPI(system, task, m, horizon=H, num_path=N, sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    ctl.reset()
    traj = simulate(ctl, task.get_init_obs(), task.term_cond, sim_model=sur, max_steps=T)
    assert rel_err(traj.obs, g[tag + "_obs"]) < 1e-8 and rel_err(traj.ctrls, g[tag + "_ctrls"]) < 1e-8
    # device loop, replaying the draws eval_cfg consumed: (H,1) at construction, (H,1) at reset, (N,H,1) per step
    np.random.seed(int(g[tag + "_np_seed"]))
    np.random.normal(scale=scale, size=(H, 1))
    act0 = np.random.normal(scale=scale, size=(H, 1))
    eps = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(T - 1)])
    ev = CandidateEvaluator(system, task, m, surrogate=sur)
    cand = dict(horizon=H, sigma=float(g["sigma"]), lmda=float(g["lmda"]), num_path=N, Q=g["Q"], R=g["R"], F=g["F"])
    scores, obs, ctrls = ev.evaluate([cand], eps_all=eps, act_init=act0, return_trajectories=True)
    assert obs.shape == (1, T, sur.state_dim)
    assert rel_err(obs[0][:, :3], g[tag + "_obs"]) < 1e-8 and rel_err(ctrls[0], g[tag + "_ctrls"]) < 1e-8
    assert abs(scores[0] - g[tag + "_cost"]) < 1e-7 * abs(g[tag + "_cost"])
    # a batch of lifted candidates: each equals itself alone (Philox noise keyed by the global index)
    c2 = dict(cand, horizon=6, num_path=100, sigma=0.9)
    both = ev.evaluate([cand, c2, cand], seed=5)
    assert both[1] == ev.evaluate([c2], seed=5, index_offset=1)[0]


# ---- iLQR beyond 64 model states (ARX history 4 on 18 observations / 6 controls: 91 states) ------------
def _arx4(g):
    from autompc_amd import ARX
    system = make_system(18, 6, dt=float(g["dt"]))
    m = ARX(system, history=4)
    m.set_parameters({"coeffs": g["coeffs"]})
    assert m.state_dim == int(g["state_dim"]) == 91
    assert abs(m.A.sum() - g["A_sum"]) < 1e-9 * g["A_abs_sum"] and abs(np.abs(m.B).sum() - g["B_abs_sum"]) < 1e-12 * g["B_abs_sum"]
    return system, m


@pytest.mark.parametrize("tag", ["free", "clip"])
def test_oracle_ilqr_on_the_91_state_arx_matches_reference(tag):
    """The reference's compute_ilqr_default (ilqr.py:100-265) on its ARX model with the default history 4
    (arx.py:27,37-45): pins the oracle beyond 64 states."""
    from oracle.ilqr import ILQROracle
    g = golden("wideilqr_arx4_hc")
    system, m = _arx4(g)
    orc_model = ARXOracle(system, 4, m.A, m.B)
    ub = (np.full(6, g["clip_bounds"][0]), np.full(6, g["clip_bounds"][1])) if tag == "clip" else None
    orc = ILQROracle(orc_model, QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"]), float(g["dt"]), int(g["H"]), ubounds=ub)
    conv, st, ct, Ks, ks = orc.solve(g["x0"], np.zeros((int(g["H"]), 6)))
    assert conv == bool(g[tag + "_converged"])
    assert rel_err(st, g[tag + "_states"]) < 1e-6 and rel_err(ct, g[tag + "_ctrls"]) < 1e-6
    assert rel_err(Ks, g[tag + "_Ks"]) < 1e-5 and np.max(np.abs(ks - g[tag + "_ks"])) < 1e-9 * max(1.0, np.max(np.abs(ct)))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["free", "clip"])
def test_device_ilqr_on_the_91_state_arx_matches_reference(tag):
    """The drop-in IterativeLQR on the reference's default-history ARX model (91 states: ilqr_wide.hpp):
    compute_ilqr_default and run() against the reference's own results."""
    from autompc_amd import IterativeLQR, QuadCost, Task
    g = golden("wideilqr_arx4_hc")
    system, m = _arx4(g)
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    if tag == "clip":
        task.set_ctrl_bounds(np.full(6, g["clip_bounds"][0]), np.full(6, g["clip_bounds"][1]))
    ctl = IterativeLQR(system, task, m, int(g["H"]))
    conv, st, ct, Ks, ks = ctl.compute_ilqr_default(g["x0"], np.zeros((int(g["H"]), 6)))
    assert conv == bool(g[tag + "_converged"])
    assert rel_err(st, g[tag + "_states"]) < 1e-6 and rel_err(ct, g[tag + "_ctrls"]) < 1e-6
    # (at convergence the feed-forward terms are rounding noise, ~1e-15: an absolute bound on the controls' scale)
    assert rel_err(Ks, g[tag + "_Ks"]) < 1e-5 and np.max(np.abs(ks - g[tag + "_ks"])) < 1e-9 * max(1.0, np.max(np.abs(ct)))
    u, newstate = ctl.run(np.concatenate([g["x0"], np.zeros(6)]), g["run_obs"])
    assert rel_err(u, g[tag + "_u"]) < 1e-6 and rel_err(newstate, g[tag + "_newstate"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # ns, nu, no, H, bounded, dense cost
    (66, 6, 18, 12, False, False),          # ARX history 3 on HalfCheetah: just past the MLP tile's 64
    (115, 6, 18, 10, True, True),           # history 5
    (128, 8, 20, 8, False, True),           # the largest shape
    (97, 1, 4, 15, True, False),
])
def test_wide_linear_ilqr_vs_oracle_and_queue(case):
    """Random stable linear models: the device solve against the oracle, and problems of different
    horizons / cost blocks streamed through one plan (ampc_ilqr_solve_queue_var) equal to one-problem solves."""
    from autompc_amd import _lib
    from oracle.ilqr import ILQROracle
    ns, nu, no, H, bounded, dense = case
    rng = np.random.default_rng(ns + nu)
    A = 0.92 * np.linalg.qr(rng.normal(size=(ns, ns)))[0] + 0.02 * rng.normal(size=(ns, ns)) / np.sqrt(ns)
    Bm = rng.normal(scale=0.3, size=(ns, nu))
    system = make_system(no, nu)

    class Lin:
        state_dim = ns

        def __init__(self):
            self.system = system

        def pred(self, s, u):
            return A @ s + Bm @ u

        def pred_batch(self, s, u):
            return s @ A.T + u @ Bm.T

        def pred_diff(self, s, u):
            return A @ s + Bm @ u, A.copy(), Bm.copy()

        def pred_diff_batch(self, s, u):
            n = s.shape[0]
            return s @ A.T + u @ Bm.T, np.tile(A, (n, 1, 1)), np.tile(Bm, (n, 1, 1))
    C = 3
    costs = []
    for c in range(C):
        if dense:
            W = rng.normal(size=(no, no))
            Q, F = W @ W.T / no + 0.05 * rng.normal(size=(no, no)), np.diag(rng.uniform(0.5, 2, size=no))
            R = np.diag(rng.uniform(0.05, 0.2, size=nu)) + 0.01
        else:
            Q, F, R = np.diag(rng.uniform(0.5, 2, size=no)), np.diag(rng.uniform(0.5, 2, size=no)), np.diag(rng.uniform(0.05, 0.2, size=nu))
        costs.append((Q, R, F, rng.normal(scale=0.1, size=no)))
    h = _lib.Handle(0, "f64")
    h.set_linear(A, Bm)
    h.set_quad_costs(np.stack([c[0] for c in costs]), np.stack([c[1] for c in costs]), np.stack([c[2] for c in costs]),
                     np.stack([c[3] for c in costs]))
    lo, hi = -0.3, 0.4
    if bounded:
        h.set_ctrl_bounds(np.full(nu, lo), np.full(nu, hi))
    P = 7
    x0 = rng.uniform(-0.5, 0.5, size=(P, ns))
    ci = rng.integers(0, C, size=P).astype(np.int32)
    hz = rng.integers(3, H + 1, size=P).astype(np.int32)
    hz[0] = H
    model = Lin()
    for j in range(3):
        Q, R, F, goal = costs[ci[j]]
        orc = ILQROracle(model, QuadCostOracle(Q, R, F, goal), 0.05, int(hz[j]),
                         ubounds=(np.full(nu, lo), np.full(nu, hi)) if bounded else None)
        conv, st, ct, Ks, ks = orc.solve(x0[j], np.zeros((int(hz[j]), nu)))
        one = _lib.IlqrPlan(h, 1, int(hz[j]), 0.05, cost_index=ci[j:j + 1], clip_to_bounds=bounded)
        got = one.solve(x0[j], np.zeros((int(hz[j]), nu)), max_iter=50)
        one.close()
        assert bool(got["converged"][0]) == conv and int(got["iters"][0]) == orc.n_iter
        assert rel_err(got["states"][0], st) < 1e-7 and rel_err(got["ctrls"][0], ct) < 1e-6
        assert rel_err(got["Ks"][0], Ks) < 1e-6 and abs(got["objective"][0] - orc.final_obj) < 1e-8 * max(1.0, abs(orc.final_obj))
    plan = _lib.IlqrPlan(h, 3, H, 0.05, cost_index=np.zeros(3, dtype=np.int32), clip_to_bounds=bounded)
    q = plan.solve_queue(x0, None, ci, max_iter=50, horizon=hz)
    for j in range(P):
        Hj = int(hz[j])
        one = _lib.IlqrPlan(h, 1, Hj, 0.05, cost_index=ci[j:j + 1], clip_to_bounds=bounded)
        ref = one.solve(x0[j], np.zeros((Hj, nu)), max_iter=50)
        one.close()
        for k in ("converged", "iters", "status", "objective"):
            np.testing.assert_array_equal(q[k][j], ref[k][0])
        np.testing.assert_array_equal(q["states"][j, :Hj + 1], ref["states"][0])
        np.testing.assert_array_equal(q["Ks"][j, :Hj], ref["Ks"][0])
    plan.close()
    h.close()


@pytest.mark.gpu
def test_ilqr_candidate_evaluator_on_the_91_state_arx():
    """IlqrCandidateEvaluator with an ARX controller model that is its own surrogate (history 4: 91 states,
    the stacked history carried on the device from one solve to the next): device-resident episodes against
    host simulate() + the drop-in controller, and against the per-step host loop."""
    from autompc_amd import IterativeLQR, QuadCost, Task, simulate
    from autompc_amd.tuning import IlqrCandidateEvaluator
    g = golden("wideilqr_arx4_hc")
    system, m = _arx4(g)
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    task.set_ctrl_bounds(np.full(6, -0.3), np.full(6, 0.3))
    task.set_init_obs(np.random.default_rng(2).uniform(-0.3, 0.3, size=18))
    T = 6
    task.set_num_steps(T)
    rng = np.random.default_rng(5)
    cands = [dict(horizon=int(h), Q=rng.uniform(0.5, 2.0, size=18), R=rng.uniform(0.05, 0.3, size=6),
                  F=rng.uniform(0.5, 2.0, size=18)) for h in (6, 11, 8, 14)]
    dev = IlqrCandidateEvaluator(system, task, m, max_slots=3)
    sd, od, cd = dev.evaluate(cands, return_trajectories=True)
    sh, oh, ch = IlqrCandidateEvaluator(system, task, m, device_resident=False).evaluate(cands, return_trajectories=True)
    np.testing.assert_allclose(od, oh, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(sd, sh, rtol=1e-11)
    assert od.shape == (4, T, 91) and np.all(np.isfinite(sd))
    c = cands[1]
    t1 = Task(system)
    t1.set_cost(QuadCost(system, np.diag(c["Q"]), np.diag(c["R"]), np.diag(c["F"]), goal=g["goal"]))
    t1.set_ctrl_bounds(np.full(6, -0.3), np.full(6, 0.3))
    ctl = IterativeLQR(system, t1, m, c["horizon"])
    traj = simulate(ctl, task.get_init_obs(), task.term_cond, sim_model=m, max_steps=T)
    assert np.max(np.abs(od[1][:, :18] - traj.obs)) < 1e-8 and np.max(np.abs(cd[1] - traj.ctrls)) < 1e-8
    assert abs(sd[1] - task.get_cost()(traj)) < 1e-8 * abs(sd[1])
