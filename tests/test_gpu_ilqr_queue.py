"""Continuous batching of iLQR problems (ampc_ilqr_solve_queue): P problems streamed through B slots,
refilled on the device, must give every problem exactly what a one-problem ampc_ilqr_solve gives it
(IterativeLQR.compute_ilqr_default, ilqr.py:100-265).  Needs MI355X."""
import numpy as np
import pytest

from oracle import mlp as omlp
from oracle.costs import QuadCostOracle
from oracle.ilqr import ILQROracle
from oracle.mlp import MLPOracle
from helpers import make_system

pytestmark = pytest.mark.gpu

KEYS = ("states", "ctrls", "Ks", "ks", "converged", "iters", "status", "objective")


def _setup(nx, nu, hidden, act, C, seed, bounds=None, precision="f64"):
    from autompc_amd import _lib
    p = omlp.random_params(nx, nu, hidden, act, seed=seed)
    rng = np.random.default_rng(seed)
    Q = np.stack([np.diag(rng.uniform(0.5, 2.0, size=nx)) for _ in range(C)])
    R = np.stack([np.diag(rng.uniform(0.05, 0.2, size=nu)) for _ in range(C)])
    F = np.stack([np.diag(rng.uniform(0.5, 2.0, size=nx)) for _ in range(C)])
    goal = rng.normal(scale=0.05, size=(C, nx))
    h = _lib.Handle(0, precision)
    h.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    h.set_quad_costs(Q, R, F, goal)
    if bounds is not None:
        h.set_ctrl_bounds(np.full(nu, bounds[0]), np.full(nu, bounds[1]))
    return p, h, (Q, R, F, goal)


@pytest.mark.parametrize("case", [
    # nx, nu, hidden, act, H, B, P, bounds, max_iter
    (17, 6, [256, 256], "relu", 50, 6, 20, (-0.25, 0.25), 50),      # BASELINE config 4 shape: MFMA sweep + four-row search
    (17, 6, [256, 256], "tanh", 20, 4, 9, None, 12),                # the per-problem iteration cap bites
    (5, 2, [64, 48], "tanh", 15, 8, 3, None, 50),                   # fewer problems than slots; run-time shapes
    (40, 3, [64], "tanh", 10, 3, 7, None, 30),                      # wide states: general sweep + sixteen-row search
])
def test_queue_equals_one_problem_solves(case):
    from autompc_amd import _lib
    nx, nu, hidden, act, H, B, P, bounds, max_iter = case
    C = 5
    p, h, _ = _setup(nx, nu, hidden, act, C, seed=nx + H, bounds=bounds)
    rng = np.random.default_rng(P)
    x0 = rng.uniform(-0.2, 0.2, size=(P, nx))
    ug = np.zeros((P, H, nu))
    ug[::2] = rng.uniform(-0.05, 0.05, size=ug[::2].shape)          # some problems start from a non-zero guess
    ci = rng.integers(0, C, size=P).astype(np.int32)
    plan = _lib.IlqrPlan(h, B, H, 0.05, cost_index=np.zeros(B, dtype=np.int32), clip_to_bounds=bounds is not None)
    got = plan.solve_queue(x0, ug, ci, max_iter=max_iter)
    assert got["iters"].max() <= max_iter
    one = _lib.IlqrPlan(h, 1, H, 0.05, cost_index=np.zeros(1, dtype=np.int32), clip_to_bounds=bounds is not None)
    for j in range(P):
        one.close()
        one = _lib.IlqrPlan(h, 1, H, 0.05, cost_index=ci[j:j + 1], clip_to_bounds=bounds is not None)
        ref = one.solve(x0[j], ug[j], max_iter=max_iter)
        for k in KEYS:
            np.testing.assert_array_equal(got[k][j], ref[k][0], err_msg="problem %d, %s" % (j, k))
    # the plan is an ordinary plan again afterwards, and a second queue run gives the same answers
    again = plan.solve_queue(x0, ug, ci, max_iter=max_iter)
    for k in KEYS:
        np.testing.assert_array_equal(again[k], got[k])
    ordinary = plan.solve(x0[:B] if P >= B else np.tile(x0[:1], (B, 1)), np.zeros((B, H, nu)), max_iter=3)
    assert ordinary["iters"].max() <= 3
    assert plan.stats()["candidate_rows"] > 0


def test_queue_against_the_oracle_and_failure_isolation():
    """Queue results against independent oracle solves, with a singular problem in the middle of the
    queue: it reports status 1 (the reference's LinAlgError) and nobody else notices."""
    from autompc_amd import _lib
    nx, nu, H, B, P = 4, 2, 12, 3, 8
    p, h, (Q, R, F, goal) = _setup(nx, nu, [64, 64], "tanh", 2, seed=3)
    # cost block 1: R = 0 on a model... keep the model, make Quu singular through R = 0 and F = Q = 0
    Q2, R2, F2 = Q.copy(), R.copy(), F.copy()
    Q2[1], R2[1], F2[1] = 0.0, 0.0, 0.0
    h.set_quad_costs(Q2, R2, F2, goal)
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-0.3, 0.3, size=(P, nx))
    ci = np.zeros(P, dtype=np.int32)
    ci[3] = 1
    plan = _lib.IlqrPlan(h, B, H, 0.05)
    got = plan.solve_queue(x0, None, ci, max_iter=50)
    assert got["status"][3] == 1 and (np.delete(got["status"], 3) == 0).all()
    system = make_system(nx, nu, dt=0.05)
    for j in range(P):
        if j == 3:
            continue
        orc = ILQROracle(MLPOracle(system, p), QuadCostOracle(Q2[0], R2[0], F2[0], goal[0]), 0.05, H)
        conv, st, ct, Ks, ks = orc.solve(x0[j], np.zeros((H, nu)))
        assert bool(got["converged"][j]) == conv and int(got["iters"][j]) == orc.n_iter
        assert np.max(np.abs(got["states"][j] - st)) < 1e-6 * max(1.0, np.max(np.abs(st)))
        assert abs(got["objective"][j] - orc.final_obj) < 1e-8 * max(1.0, abs(orc.final_obj))


def test_device_resident_ilqr_episodes_equal_the_host_loop():
    """ampc_ilqr_closed_loop: whole episodes (solve -> surrogate step -> next solve) on the device, the
    candidates streaming through fewer slots than there are candidates, against the per-control-step
    host loop (one batched solve + one surrogate step per step) and host simulate() + the drop-in
    controller; a candidate whose Quu is singular stops with `failed` and scores inf."""
    from autompc_amd import MLP, IterativeLQR, QuadCost, Task, simulate
    from autompc_amd.tuning import IlqrCandidateEvaluator, random_ilqr_candidates
    nx, nu, T = 4, 2, 9
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [64, 48], "tanh", seed=21)

    def model_of(pp):
        m = MLP(system, n_hidden_layers=2, hidden_size_1=64, hidden_size_2=48, nonlintype="tanh")
        m.weights, m.biases = [w.copy() for w in pp["weights"]], [b.copy() for b in pp["biases"]]
        m.xu_means, m.xu_std, m.dy_means, m.dy_std = pp["xu_means"], pp["xu_std"], pp["dy_means"], pp["dy_std"]
        return m
    model = model_of(p)
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), 2.0 * np.eye(nx)))
    task.set_ctrl_bounds(-0.6 * np.ones(nu), 0.6 * np.ones(nu))
    task.set_init_obs(np.array([0.3, -0.2, 0.25, 0.1]))
    task.set_num_steps(T)
    cands = random_ilqr_candidates(system, 10, seed=4)
    for c in cands:
        c["Q"], c["R"], c["F"] = c["Q"] ** 0.3, c["R"] ** 0.3, c["F"] ** 0.3
        c["horizon"] = 8 if c["horizon"] % 2 else 12                 # two horizon groups of several candidates
    dev = IlqrCandidateEvaluator(system, task, model, max_slots=3)   # fewer slots than candidates per group
    host = IlqrCandidateEvaluator(system, task, model, device_resident=False)
    sd, od, cd = dev.evaluate(cands, return_trajectories=True)
    sh, oh, ch = host.evaluate(cands, return_trajectories=True)
    assert dev.last_iterations.min() >= T - 1
    np.testing.assert_allclose(od, oh, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(cd, ch, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(sd, sh, rtol=1e-12)
    c = cands[3]
    t1 = Task(system)
    t1.set_cost(QuadCost(system, np.diag(c["Q"]), np.diag(c["R"]), np.diag(c["F"])))
    t1.set_ctrl_bounds(-0.6 * np.ones(nu), 0.6 * np.ones(nu))
    ctl = IterativeLQR(system, t1, model, c["horizon"])
    traj = simulate(ctl, task.get_init_obs(), task.term_cond, sim_model=model, max_steps=T)
    assert np.max(np.abs(od[3] - traj.obs)) < 1e-9 and np.max(np.abs(cd[3] - traj.ctrls)) < 1e-9
    # singular Quu in the middle of the queue
    p2 = {k: ([w.copy() for w in v] if isinstance(v, list) else v) for k, v in p.items()}
    p2["weights"][0][:, nx:] = 0.0
    ev2 = IlqrCandidateEvaluator(system, task, model_of(p2), max_slots=2)
    good = dict(horizon=8, Q=np.ones(nx), R=0.1 * np.ones(nu), F=np.ones(nx))
    sing = dict(horizon=8, Q=np.ones(nx), R=np.zeros(nu), F=np.ones(nx))
    s2 = ev2.evaluate([good, sing, good, good])
    assert np.isinf(s2[1]) and np.all(np.isfinite(np.delete(s2, 1))) and s2[0] == s2[2] == s2[3]


def test_concurrent_queues_with_side_by_side_passes_equal_one_problem_solves():
    """Many small plans at once, each on its own handle / stream / host thread (what the candidate
    evaluator's horizon groups do), every one with so few slots that the passes of a line search run as
    separate workgroups of one launch (grid.y, ilqr_ls4.hpp): workgroup dispatch is oversubscribed and
    a pass workgroup may start after the workgroup that rolled out a slot's fresh guess has finished.
    A slot's mode must not change inside that launch (ilqr_slot_started, ilqr_kernels.hpp): every
    problem gets exactly the one-problem solve's result, every time."""
    from concurrent.futures import ThreadPoolExecutor
    from autompc_amd import _lib
    nx, nu, hidden, act, H, B, P, max_iter, n_plans = 17, 6, [256, 256], "relu", 12, 3, 14, 20, 12
    items = []
    for g in range(n_plans):
        p, h, _ = _setup(nx, nu, hidden, act, 3, seed=100 + g, bounds=(-0.25, 0.25))
        rng = np.random.default_rng(g)
        x0 = rng.uniform(-0.2, 0.2, size=(P, nx))
        ci = rng.integers(0, 3, size=P).astype(np.int32)
        plan = _lib.IlqrPlan(h, B, H, 0.05, cost_index=np.zeros(B, dtype=np.int32), clip_to_bounds=True)
        items.append((h, plan, x0, ci))
    refs = []
    for h, plan, x0, ci in items:
        rows = []
        for j in range(P):
            one = _lib.IlqrPlan(h, 1, H, 0.05, cost_index=ci[j:j + 1], clip_to_bounds=True)
            rows.append(one.solve(x0[j], np.zeros((H, nu)), max_iter=max_iter))
            one.close()
        refs.append(rows)

    def run(item):
        _, plan, x0, ci = item
        return plan.solve_queue(x0, None, ci, max_iter=max_iter)
    for rep in range(6):
        with ThreadPoolExecutor(max_workers=n_plans) as pool:
            outs = list(pool.map(run, items))
        for g, got in enumerate(outs):
            for j in range(P):
                for k in KEYS:
                    np.testing.assert_array_equal(got[k][j], refs[g][j][k][0],
                                                  err_msg="round %d, plan %d, problem %d, %s" % (rep, g, j, k))
    for h, plan, _, _ in items:
        plan.close()
        h.close()


@pytest.mark.parametrize("case", [
    # nx, nu, hidden, act, Hmax, B, P, bounds, max_iter
    (17, 6, [256, 256], "relu", 25, 6, 30, (-0.25, 0.25), 50),     # the tuner's range 5..25 on the c4 shape (MFMA sweep, four-row search)
    (17, 6, [256, 256], "tanh", 25, 200, 260, None, 15),            # many slots: the twelve-row search takes over
    (5, 2, [64, 48], "tanh", 18, 4, 11, None, 50),                  # run-time shapes
    (40, 3, [64], "tanh", 12, 3, 7, None, 30),                      # wide states: general sweep + sixteen-row search
])
def test_one_plan_for_every_horizon_equals_one_horizon_plans(case):
    """ampc_ilqr_solve_queue_var: problems of different horizons (IterativeLQRFactory's 5..25,
    control/ilqr.py:31-41) stream through ONE plan built for the longest; each gets bit for bit what a
    one-problem plan of its own horizon gives, and rows past its horizon come back zero."""
    from autompc_amd import _lib
    nx, nu, hidden, act, Hmax, B, P, bounds, max_iter = case
    C = 4
    p, h, _ = _setup(nx, nu, hidden, act, C, seed=nx + Hmax, bounds=bounds)
    rng = np.random.default_rng(P)
    x0 = rng.uniform(-0.2, 0.2, size=(P, nx))
    hz = rng.integers(max(2, Hmax // 5), Hmax + 1, size=P).astype(np.int32)
    hz[0], hz[-1] = Hmax, max(2, Hmax // 5)
    ug = np.zeros((P, Hmax, nu))
    ug[::3] = rng.uniform(-0.05, 0.05, size=ug[::3].shape)
    ci = rng.integers(0, C, size=P).astype(np.int32)
    plan = _lib.IlqrPlan(h, B, Hmax, 0.05, cost_index=np.zeros(B, dtype=np.int32), clip_to_bounds=bounds is not None)
    got = plan.solve_queue(x0, ug, ci, max_iter=max_iter, horizon=hz)
    check = range(P) if P <= 40 else rng.choice(P, size=40, replace=False)
    for j in check:
        H = int(hz[j])
        one = _lib.IlqrPlan(h, 1, H, 0.05, cost_index=ci[j:j + 1], clip_to_bounds=bounds is not None)
        ref = one.solve(x0[j], ug[j, :H], max_iter=max_iter)
        one.close()
        for k in ("converged", "iters", "status", "objective"):
            np.testing.assert_array_equal(got[k][j], ref[k][0], err_msg="problem %d (H %d), %s" % (j, H, k))
        np.testing.assert_array_equal(got["states"][j, :H + 1], ref["states"][0], err_msg="problem %d states" % j)
        for k in ("ctrls", "Ks", "ks"):
            np.testing.assert_array_equal(got[k][j, :H], ref[k][0], err_msg="problem %d %s" % (j, k))
            assert not got[k][j, H:].any()
        assert not got["states"][j, H + 1:].any()
    # the same plan without horizons is an ordinary queue again
    plain = plan.solve_queue(x0[:B], None, ci[:B], max_iter=3)
    one = _lib.IlqrPlan(h, 1, Hmax, 0.05, cost_index=ci[:1], clip_to_bounds=bounds is not None)
    ref = one.solve(x0[0], np.zeros((Hmax, nu)), max_iter=3)
    np.testing.assert_array_equal(plain["states"][0], ref["states"][0])
    with pytest.raises(ValueError):
        plan.solve_queue(x0, None, ci, horizon=np.full(P, Hmax + 1))


def test_evaluator_one_plan_equals_horizon_groups():
    """IlqrCandidateEvaluator: all horizons through one plan (device-resident episodes and the host loop)
    against the round-4 scheme of one plan per horizon -- the same scores and trajectories bit for bit."""
    from autompc_amd import MLP, QuadCost, Task
    from autompc_amd.tuning import IlqrCandidateEvaluator, random_ilqr_candidates
    nx, nu, T = 4, 2, 7
    system = make_system(nx, nu)
    pp = omlp.random_params(nx, nu, [64, 48], "tanh", seed=21)
    m = MLP(system, n_hidden_layers=2, hidden_size_1=64, hidden_size_2=48, nonlintype="tanh")
    m.weights, m.biases = [w.copy() for w in pp["weights"]], [b.copy() for b in pp["biases"]]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = pp["xu_means"], pp["xu_std"], pp["dy_means"], pp["dy_std"]
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), 2.0 * np.eye(nx)))
    task.set_ctrl_bounds(-0.6 * np.ones(nu), 0.6 * np.ones(nu))
    task.set_init_obs(np.array([0.3, -0.2, 0.25, 0.1]))
    task.set_num_steps(T)
    cands = random_ilqr_candidates(system, 14, seed=9)
    for c in cands:
        c["Q"], c["R"], c["F"] = c["Q"] ** 0.3, c["R"] ** 0.3, c["F"] ** 0.3
    for kw in ({"max_slots": 5}, {}, {"device_resident": False}, {"device_resident": False, "max_slots": 4}):
        mode = {"device_resident": kw.get("device_resident", True)}
        ref = IlqrCandidateEvaluator(system, task, m, one_plan=False, **mode).evaluate(cands, return_trajectories=True)
        got = IlqrCandidateEvaluator(system, task, m, **kw).evaluate(cands, return_trajectories=True)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b, err_msg=str(kw))


@pytest.mark.parametrize("case", [
    # nx, nu, hidden, act, H, B, P, bounds, max_iter, horizons
    (17, 6, [256, 256], "relu", 20, 300, 300, (-0.25, 0.25), 14, False),   # all admitted at once, twelve-row search
    (5, 2, [64, 48], "tanh", 12, 280, 700, None, 20, True),                # refills + per-problem horizons
])
def test_more_slots_than_cus_take_the_slots_with_work_first(case):
    """A plan with more slots than the GPU has CUs (round 5): every kernel of an iteration takes its slot from
    the iteration's live-first order (ilqr_compact_kernel) and the grids shrink with the polled count of slots
    that still have work -- each problem still gets bit for bit what a one-problem solve gives it."""
    from autompc_amd import _lib
    nx, nu, hidden, act, H, B, P, bounds, max_iter, var_h = case
    C = 3
    p, h, _ = _setup(nx, nu, hidden, act, C, seed=nx + H, bounds=bounds)
    rng = np.random.default_rng(P)
    x0 = rng.uniform(-0.2, 0.2, size=(P, nx))
    ci = rng.integers(0, C, size=P).astype(np.int32)
    hz = rng.integers(3, H + 1, size=P).astype(np.int32) if var_h else None
    plan = _lib.IlqrPlan(h, B, H, 0.05, cost_index=np.zeros(B, dtype=np.int32), clip_to_bounds=bounds is not None)
    got = plan.solve_queue(x0, None, ci, max_iter=max_iter, horizon=hz)
    assert len(set(got["iters"].tolist())) > 3          # problems finish at different times: the live set thins out
    for j in rng.choice(P, size=24, replace=False):
        Hj = int(hz[j]) if var_h else H
        one = _lib.IlqrPlan(h, 1, Hj, 0.05, cost_index=ci[j:j + 1], clip_to_bounds=bounds is not None)
        ref = one.solve(x0[j], np.zeros((Hj, nu)), max_iter=max_iter)
        one.close()
        for k in ("converged", "iters", "status", "objective"):
            np.testing.assert_array_equal(got[k][j], ref[k][0], err_msg="problem %d, %s" % (j, k))
        np.testing.assert_array_equal(got["states"][j, :Hj + 1], ref["states"][0])
        for k in ("ctrls", "Ks", "ks"):
            np.testing.assert_array_equal(got[k][j, :Hj], ref[k][0], err_msg="problem %d, %s" % (j, k))
    plan.close()
    h.close()
