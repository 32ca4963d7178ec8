"""Controller costs that are sums of quadratic terms with different goals (QuadCostFactory +
GaussRegFactory, gauss_reg_factory.py:37-45; SumCost._sum_results, sum_cost.py:49-54) on the device:
the reference evaluates them term by term (mppi.py:73-82, ilqr.py:124-129,159-174), the kernels as one
affine-quadratic block (ampc_set_affine_quad_costs).  Needs MI355X.  The per-solve goldens
(mppi_sumcost_*.npz, ilqr_sumcost_*.npz) run in test_gpu_mppi.py / test_gpu_ilqr.py with the rest."""
import numpy as np
import pytest

from conftest import golden
from helpers import check_weights, golden_params, hip_cost_from_golden, make_system, rel_err
from oracle import mlp as omlp
from oracle.costs import SumCostOracle
from oracle.ilqr import ILQROracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle

pytestmark = pytest.mark.gpu


def _hip_model(system, p, precision="f64"):
    from autompc_amd import MLP
    m = MLP(system, n_hidden_layers=len(p["weights"]) - 1, nonlintype=p["activation"], precision=precision,
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    return m


def _stack(g, bounded):
    """(system, p, scoring task, controller task): the tuner scores with the task's own cost, the
    controller is handed a task carrying the factories' cost (pipeline.py:156-160)."""
    from autompc_amd import QuadCost, Task
    nx = int(g["nx"])
    system = make_system(nx, 1)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    task, ctask = Task(system), Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    ctask.set_cost(hip_cost_from_golden(system, g))
    for t in (task, ctask):
        if bounded:
            t.set_ctrl_bounds([g["bounds"][0]], [g["bounds"][1]])
        t.set_init_obs(g["init"])
        t.set_num_steps(int(g["num_steps"]))
    return system, p, task, ctask


def test_mppi_evalcfg_with_a_sum_cost_drop_in_and_evaluator():
    from autompc_amd import MPPI, simulate
    from autompc_amd.tuning import CandidateEvaluator
    g = golden("loop_evalcfg_sumcost")
    system, p, task, ctask = _stack(g, True)
    model = _hip_model(system, p)
    # the drop-in controller in the default (numpy) noise mode, driven as eval_cfg drives it
    np.random.seed(int(g["np_seed"]))
    ctl = MPPI(system, ctask, model, horizon=int(g["H"]), num_path=int(g["N"]), sigma=float(g["sigma"]),
               lmda=float(g["lmda"]))
    ctl.reset()
    traj = simulate(ctl, task.get_init_obs(), task.term_cond, sim_model=model, max_steps=task.get_num_steps())
    assert rel_err(traj.obs, g["surr_obs"]) < 1e-7 and rel_err(traj.ctrls, g["surr_ctrls"]) < 1e-7
    assert abs(task.get_cost()(traj) - g["surr_cost"]) < 1e-7 * abs(g["surr_cost"])
    assert abs(ctask.get_cost()(traj) - g["ctl_cost_of_traj"]) < 1e-7 * abs(g["ctl_cost_of_traj"])
    # the batched evaluator: the candidate carries its controller cost as an object
    N, H, scale = int(g["N"]), int(g["H"]), np.sqrt(float(g["sigma"]))
    np.random.seed(int(g["np_seed"]))
    np.random.normal(scale=scale, size=(H, 1))
    act0 = np.random.normal(scale=scale, size=(H, 1))
    eps = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(int(g["num_steps"]) - 1)])
    ev = CandidateEvaluator(system, task, model)
    cand = dict(horizon=H, sigma=float(g["sigma"]), lmda=float(g["lmda"]), num_path=N, cost=ctask.get_cost())
    scores, obs, ctrls = ev.evaluate([cand], eps_all=eps, act_init=act0, return_trajectories=True)
    assert rel_err(obs[0], g["surr_obs"]) < 1e-7 and rel_err(ctrls[0], g["surr_ctrls"]) < 1e-7
    assert abs(scores[0] - g["surr_cost"]) < 1e-7 * abs(g["surr_cost"])
    # a sum-cost candidate next to plain ones in one plan leaves everybody's rows unchanged
    plain = dict(horizon=H, sigma=float(g["sigma"]), lmda=float(g["lmda"]), num_path=N, Q=g["Q"], R=g["R"], F=g["F"])
    both = ev.evaluate([plain, cand, plain], seed=3)
    alone = ev.evaluate([cand], seed=3, index_offset=1)
    assert both[1] == alone[0]


def test_ilqr_evalcfg_with_a_sum_cost_drop_in_and_evaluator():
    from autompc_amd import IterativeLQR, simulate
    from autompc_amd.tuning import IlqrCandidateEvaluator
    g = golden("loop_evalcfg_sumcost_ilqr")
    system, p, task, ctask = _stack(g, False)
    model = _hip_model(system, p)
    ctl = IterativeLQR(system, ctask, model, int(g["H"]))
    ctl.reset()
    traj = simulate(ctl, task.get_init_obs(), task.term_cond, sim_model=model, max_steps=task.get_num_steps())
    assert rel_err(traj.obs, g["surr_obs"]) < 1e-6 and rel_err(traj.ctrls, g["surr_ctrls"]) < 1e-6
    assert abs(task.get_cost()(traj) - g["surr_cost"]) < 1e-6 * abs(g["surr_cost"])
    ev = IlqrCandidateEvaluator(system, task, model)
    cand = dict(horizon=int(g["H"]), cost=ctask.get_cost())
    scores, obs, ctrls = ev.evaluate([cand], return_trajectories=True)
    assert rel_err(obs[0], g["surr_obs"]) < 1e-6 and rel_err(ctrls[0], g["surr_ctrls"]) < 1e-6
    assert abs(scores[0] - g["surr_cost"]) < 1e-6 * abs(g["surr_cost"])


def _random_sum(system, rng, n_terms, diag):
    from autompc_amd import QuadCost
    no, nu = system.obs_dim, system.ctrl_dim
    terms, arrs = [], []
    for _ in range(n_terms):
        if diag:
            Q, F = np.diag(rng.uniform(0.2, 2.0, size=no)), np.diag(rng.uniform(0.2, 2.0, size=no))
            R = np.diag(rng.uniform(0.01, 0.1, size=nu))
        else:
            A, B = rng.normal(size=(no, no)), rng.normal(size=(no, no))
            Q, F = A @ A.T / no + 0.1 * rng.normal(size=(no, no)), B @ B.T / no
            R = np.diag(rng.uniform(0.01, 0.1, size=nu)) + 0.003 * rng.normal(size=(nu, nu))
        goal = rng.normal(scale=0.3, size=no)
        terms.append(QuadCost(system, Q, R, F, goal=goal))
        arrs.append((Q, R, F, goal))
    cost = terms[0]
    for t in terms[1:]:
        cost = cost + t
    return cost, SumCostOracle.from_arrays(*zip(*arrs))


@pytest.mark.parametrize("shape", [
    # nx, nu, hidden, act, N, H, diag cost, per-particle terminal
    (17, 6, [256, 256], "relu", 4096, 30, True, False),      # BASELINE config 3, sixteen-row kernel, diagonal path
    (17, 6, [256, 256], "relu", 512, 12, False, True),       # dense path, per-particle terminal
    (2, 1, [64, 64], "relu", 1024, 30, True, False),         # BASELINE config 2, four-row kernel
    (5, 3, [100, 40], "tanh", 77, 9, False, False),          # run-time-shape kernels, ragged tile
    (3, 2, [48], "selu", 200, 7, True, True),
])
def test_mppi_sum_cost_vs_oracle(shape):
    """Every rollout kernel (sixteen-row static / run-time shape, four-row) and both cost paths
    (diagonal: affine part fused into the state update; dense: row-parallel) against the oracle's
    term-by-term evaluation, at the full BASELINE sizes too."""
    from autompc_amd import MPPI, Task
    nx, nu, hidden, act, N, H, diag, ppt = shape
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, hidden, act, seed=nx + 40)
    rng = np.random.default_rng(nx * 7 + N)
    cost, ocost = _random_sum(system, rng, 3, diag)
    task = Task(system)
    task.set_cost(cost)
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    omodel = MLPOracle(system, p)
    np.random.seed(5)
    orc = MPPIOracle(omodel, ocost, np.tile([-1.0, 1.0], (nu, 1)), horizon=H, num_path=N, sigma=0.9, lmda=0.8,
                     per_particle_terminal=ppt)
    np.random.seed(5)
    ctl = MPPI(system, task, _hip_model(system, p), horizon=H, num_path=N, sigma=0.9, lmda=0.8,
               per_particle_terminal=ppt)
    obs = rng.uniform(-0.1, 0.1, size=nx)
    cs = np.concatenate([obs, np.zeros(nu)])
    state = np.random.get_state()
    uo, _ = orc.run(cs, obs)
    np.random.set_state(state)
    uh, _ = ctl.run(cs, obs, return_details=True)
    assert rel_err(ctl.last_costs, orc.last_costs) < 1e-9
    assert rel_err(ctl.act_sequence, orc.act_sequence) < 1e-8 and rel_err(uh, uo) < 1e-8


@pytest.mark.parametrize("shape", [
    # nx, nu, hidden, act, H, bounds, diag, strict_reference
    (17, 6, [256, 256], "relu", 50, None, True, True),        # BASELINE config 4: MFMA sweep + four-row line search
    (17, 6, [256, 256], "tanh", 30, (-0.3, 0.3), False, True),
    (4, 2, [64, 48], "tanh", 15, None, False, False),         # run-time shapes; terminal gradient about the goals
    (40, 3, [64], "tanh", 10, None, True, True),              # wide states: general sweep + sixteen-row line search
])
def test_ilqr_sum_cost_vs_oracle(shape):
    from autompc_amd import IterativeLQR, QuadCost, Task
    nx, nu, hidden, act, H, bounds, diag, strict = shape
    system = make_system(nx, nu, dt=0.05)
    p = omlp.random_params(nx, nu, hidden, act, seed=nx + 60)
    rng = np.random.default_rng(nx * 3 + H)
    cost, ocost = _random_sum(system, rng, 2, diag)
    if not strict:
        cost = cost.costs[0].__class__(system, *cost.costs[0].get_cost_matrices(), goal=cost.costs[0].get_goal(),
                                       strict_reference=False) + \
            QuadCost(system, *cost.costs[1].get_cost_matrices(), goal=cost.costs[1].get_goal(), strict_reference=False)

        class NonStrict(SumCostOracle):            # every term differentiates (x - g_k)'F_k(x - g_k)
            def eval_term_obs_cost_hess(self, obs):
                c = sum((obs - t.goal) @ t.F @ (obs - t.goal) for t in self.terms)
                return c, sum((t.F + t.F.T) @ (obs - t.goal) for t in self.terms), sum(t.F + t.F.T for t in self.terms)
        ocost = NonStrict(ocost.terms)
    task = Task(system)
    task.set_cost(cost)
    ub = None
    if bounds is not None:
        task.set_ctrl_bounds(np.full(nu, bounds[0]), np.full(nu, bounds[1]))
        ub = (np.full(nu, bounds[0]), np.full(nu, bounds[1]))
    ctl = IterativeLQR(system, task, _hip_model(system, p), H)
    x0 = rng.uniform(-0.2, 0.2, size=nx)
    conv, states, ctrls, Ks, ks = ctl.compute_ilqr_default(x0, np.zeros((H, nu)))
    orc = ILQROracle(MLPOracle(system, p), ocost, 0.05, H, ubounds=ub)
    oconv, ost, oct_, oKs, oks = orc.solve(x0, np.zeros((H, nu)))
    assert conv == oconv and ctl.last_iters == orc.n_iter
    assert rel_err(states, ost) < 1e-6 and rel_err(ctrls, oct_) < 1e-6
    assert rel_err(Ks, oKs) < 1e-5
    assert abs(ctl.last_objective - orc.final_obj) < 1e-8 * max(1.0, abs(orc.final_obj))


def test_sum_cost_with_a_shared_goal_is_the_plain_block_bit_for_bit():
    """lin = lin_term = consts = 0: the kernels take the same path as for one QuadCost with the summed
    matrices -- identical costs, not merely close."""
    from autompc_amd import MPPI, QuadCost, Task
    nx, nu, N, H = 17, 6, 512, 10
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=3)
    rng = np.random.default_rng(1)
    goal = rng.normal(scale=0.2, size=nx)
    Q1, Q2 = np.diag(rng.uniform(0.5, 2, size=nx)), np.diag(rng.uniform(0.5, 2, size=nx))
    R1, F1 = 0.05 * np.eye(nu), np.eye(nx)
    out = []
    for cost in (QuadCost(system, Q1 + Q2, R1 + R1, F1 + F1, goal=goal),
                 QuadCost(system, Q1, R1, F1, goal=goal) + QuadCost(system, Q2, R1, F1, goal=goal)):
        task = Task(system)
        task.set_cost(cost)
        task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
        np.random.seed(2)
        ctl = MPPI(system, task, _hip_model(system, p), horizon=H, num_path=N)
        x = np.full(nx, 0.05)
        ctl.run(np.concatenate([x, np.zeros(nu)]), x, return_details=True)
        out.append(ctl.last_costs.copy())
    np.testing.assert_array_equal(out[0], out[1])
