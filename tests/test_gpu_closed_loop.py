"""Device-resident closed loop (ampc_mppi_closed_loop) and the batched candidate evaluator vs the
reference's golden closed-loop run and the oracle (needs MI355X)."""
import numpy as np
import pytest

from conftest import golden
from helpers import check_weights, cost_from_golden, golden_params, make_system, rel_err
from oracle import mlp as omlp
from oracle.closed_loop import simulate as oracle_simulate
from oracle.costs import QuadCostOracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle

pytestmark = pytest.mark.gpu


def _hip_model(system, p, precision="f64"):
    from autompc_amd import MLP
    m = MLP(system, n_hidden_layers=len(p["weights"]) - 1, nonlintype=p["activation"],
            precision=precision,
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    return m


def test_closed_loop_matches_reference_simulate():
    """20-step simulate() of the reference (MPPI, nu = 1) reproduced with the noise stream the
    reference consumed: one (H,1) draw at construction, one (N,H,1) draw per control step."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import CandidateEvaluator
    g = golden("loop_mppi")
    nx, N, H, T = int(g["nx"]), int(g["N"]), int(g["H"]), 20
    system = make_system(nx, 1)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    task.set_ctrl_bounds([g["bounds"][0]], [g["bounds"][1]])
    np.random.seed(int(g["np_seed"]))
    scale = np.sqrt(float(g["sigma"]))
    act0 = np.random.normal(scale=scale, size=(H, 1))
    eps_all = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(T)])
    ev = CandidateEvaluator(system, task, _hip_model(system, p))
    cand = dict(horizon=H, sigma=float(g["sigma"]), lmda=float(g["lmda"]), num_path=N,
                Q=g["Q"], R=g["R"], F=g["F"])
    scores, obs, ctrls = ev.evaluate([cand], n_steps=T, init_obs=g["init"], eps_all=eps_all,
                                     act_init=act0, return_trajectories=True)
    assert rel_err(obs[0], g["obs"]) < 1e-7 and rel_err(ctrls[0], g["ctrls"]) < 1e-7
    assert abs(scores[0] - g["score"]) < 1e-7 * abs(g["score"])


def test_candidate_batch_matches_oracle_per_candidate():
    """Heterogeneous candidates (N, H, sigma, lmda, cost weights) with nu = 6 in ONE plan; a
    separately staged surrogate model; scored with the task's cost."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import CandidateEvaluator, random_candidates
    nx, nu, T = 17, 6, 6
    system = make_system(nx, nu)
    p_ctl = omlp.random_params(nx, nu, [256, 256], "relu", seed=31)
    p_sur = omlp.random_params(nx, nu, [128, 64], "tanh", seed=32)
    task = Task(system)
    Qt, Rt, Ft = np.eye(nx), 0.01 * np.eye(nu), 2.0 * np.eye(nx)
    task.set_cost(QuadCost(system, Qt, Rt, Ft))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    init = np.random.default_rng(0).uniform(-0.1, 0.1, size=nx)
    cands = random_candidates(system, 5, seed=3)
    for c in cands:                       # keep the oracle side fast and the costs O(1)
        c["num_path"] = int(c["num_path"] // 8) + 16
        c["Q"], c["R"], c["F"] = c["Q"] ** 0.1, c["R"] ** 0.1, c["F"] ** 0.1
    rng = np.random.default_rng(9)
    acts = [rng.normal(scale=np.sqrt(c["sigma"]), size=(c["horizon"], nu)) for c in cands]
    eps = [[rng.normal(scale=np.sqrt(c["sigma"]), size=(c["num_path"], c["horizon"], nu))
            for c in cands] for _ in range(T)]
    eps_all = np.concatenate([np.concatenate([e.ravel() for e in step]) for step in eps])
    ev = CandidateEvaluator(system, task, _hip_model(system, p_ctl), surrogate=_hip_model(system, p_sur))
    scores, obs, ctrls = ev.evaluate(cands, n_steps=T, init_obs=init, eps_all=eps_all,
                                     act_init=np.concatenate([a.ravel() for a in acts]),
                                     return_trajectories=True)
    sur = MLPOracle(system, p_sur)
    task_cost = QuadCostOracle(Qt, Rt, Ft, np.zeros(nx))
    for b, c in enumerate(cands):
        np.random.seed(0)
        orc = MPPIOracle(MLPOracle(system, p_ctl),
                         QuadCostOracle(np.diag(c["Q"]), np.diag(c["R"]), np.diag(c["F"]), np.zeros(nx)),
                         np.tile([-1.0, 1.0], (nu, 1)), horizon=c["horizon"], num_path=c["num_path"],
                         sigma=c["sigma"], lmda=c["lmda"])
        orc.act_sequence = acts[b].copy()
        x, cs = init.copy(), np.concatenate([init, np.zeros(nu)])
        o_obs, o_ctl = [x.copy()], []
        for s in range(T):
            u, cs = orc.run(cs, x, eps_nhu=eps[s][b])
            x = sur.pred(x, u)
            o_ctl.append(u)
            o_obs.append(x.copy())
        o_ctl.append(np.zeros(nu))
        assert rel_err(obs[b], np.array(o_obs)) < 1e-7
        assert rel_err(ctrls[b], np.array(o_ctl)) < 1e-7
        ref = task_cost.traj_cost(np.array(o_obs), np.array(o_ctl))
        assert abs(scores[b] - ref) < 1e-7 * abs(ref)


def test_c5_scale_batch_matches_oracle_per_candidate():
    """BASELINE config 5 at a size the oracle still finishes: 16 candidates drawn from the
    reference's own ranges -- horizon 5-30, num_path 100-1000, sigma, lmda (mppi.py:52-63), QuadCost
    gains log-uniform over the full 1e-3 ... 1e4 (quad_cost_factory.py:46-58), unmodified -- on the
    HalfCheetah 2 x 256 model, 50 closed-loop control steps each, one plan; every candidate's
    trajectory and score against an independent oracle closed loop fed the same noise."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import CandidateEvaluator, random_candidates
    nx, nu, T, B = 17, 6, 50, 16
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=31)
    task = Task(system)
    Qt, Rt, Ft = np.eye(nx), 0.01 * np.eye(nu), np.eye(nx)
    task.set_cost(QuadCost(system, Qt, Rt, Ft))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    init = np.random.default_rng(0).uniform(-0.1, 0.1, size=nx)
    cands = random_candidates(system, B, seed=5)
    gains = np.concatenate([np.concatenate([c["Q"], c["R"], c["F"]]) for c in cands])
    assert gains.min() < 1e-2 and gains.max() > 1e3          # the range is really exercised
    rng = np.random.default_rng(10)
    acts = [rng.normal(scale=np.sqrt(c["sigma"]), size=(c["horizon"], nu)) for c in cands]
    eps = [[rng.normal(scale=np.sqrt(c["sigma"]), size=(c["num_path"], c["horizon"], nu))
            for c in cands] for _ in range(T)]
    eps_all = np.concatenate([np.concatenate([e.ravel() for e in step]) for step in eps])
    ev = CandidateEvaluator(system, task, _hip_model(system, p))
    scores, obs, ctrls = ev.evaluate(cands, n_steps=T, init_obs=init, eps_all=eps_all,
                                     act_init=np.concatenate([a.ravel() for a in acts]),
                                     return_trajectories=True)
    model = MLPOracle(system, p)
    task_cost = QuadCostOracle(Qt, Rt, Ft, np.zeros(nx))
    worst = 0.0
    for b, c in enumerate(cands):
        np.random.seed(0)
        orc = MPPIOracle(model, QuadCostOracle(np.diag(c["Q"]), np.diag(c["R"]), np.diag(c["F"]), np.zeros(nx)),
                         np.tile([-1.0, 1.0], (nu, 1)), horizon=c["horizon"], num_path=c["num_path"],
                         sigma=c["sigma"], lmda=c["lmda"])
        orc.act_sequence = acts[b].copy()
        x, cs = init.copy(), np.concatenate([init, np.zeros(nu)])
        o_obs, o_ctl = [x.copy()], []
        for s in range(T):
            u, cs = orc.run(cs, x, eps_nhu=eps[s][b])
            x = model.pred(x, u)
            o_ctl.append(u)
            o_obs.append(x.copy())
        o_ctl.append(np.zeros(nu))
        ref = task_cost.traj_cost(np.array(o_obs), np.array(o_ctl))
        worst = max(worst, rel_err(obs[b], np.array(o_obs)), rel_err(ctrls[b], np.array(o_ctl)),
                    abs(scores[b] - ref) / abs(ref))
        # north_star: 1e-4 relative on trajectory state and cost; sharp softmin weights (costs up
        # to 1e4 x state^2 against lmda ~ 1) amplify last-bit differences over 50 closed-loop steps
        # (measured worst case over the batch: 6e-7, profiles/r03_*; asserted with margin)
        assert rel_err(obs[b], np.array(o_obs)) < 1e-5, (b, c)
        assert rel_err(ctrls[b], np.array(o_ctl)) < 1e-5, (b, c)
        assert abs(scores[b] - ref) < 1e-5 * abs(ref), (b, c)
    print("c5-scale batch: worst relative deviation from the oracle %.2e" % worst)


def test_device_noise_closed_loop_is_reproducible():
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import CandidateEvaluator
    nx, nu, T = 17, 6, 15
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=7)
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.01 * np.eye(nu), np.eye(nx)))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    task.set_init_obs(np.full(nx, 0.3))
    task.set_num_steps(T)
    ev = CandidateEvaluator(system, task, _hip_model(system, p))
    cand = dict(horizon=15, sigma=0.3, lmda=0.5, num_path=512, Q=np.ones(nx), R=0.01 * np.ones(nu),
                F=np.ones(nx))
    s1 = ev.evaluate([cand, cand], seed=11)
    s2 = ev.evaluate([cand, cand], seed=11)
    np.testing.assert_allclose(s1, s2, rtol=1e-12)
    # two identical candidates draw from different (problem-indexed) noise streams
    assert np.all(np.isfinite(s1)) and s1[0] != s1[1]
    # a different seed gives a different, equally valid, evaluation
    s3 = ev.evaluate([cand, cand], seed=12)
    assert np.all(np.isfinite(s3)) and not np.allclose(s1, s3)


def test_threshold_task_cost_is_scored_on_device():
    """Task score = quadratic + threshold + box sum (the benchmark tasks' form, e.g.
    benchmarks/cartpole.py:51): the device score of the device trajectories equals the oracle's
    term-by-term score of the same trajectories."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.costs import BoxThresholdCost, ThresholdCost, cost_terms
    from autompc_amd.tuning import CandidateEvaluator
    from oracle.costs import score_terms
    nx, nu, T = 4, 1, 25
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [64, 64], "tanh", seed=5)
    goal = np.array([0.1, 0.0, -0.1, 0.0])
    limits = np.array([[-0.5, 0.5], [-np.inf, np.inf], [-0.4, np.inf], [-2.0, 2.0]])
    cost = (ThresholdCost(system, goal, [0, 3], 0.15) + BoxThresholdCost(system, limits)
            + QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), np.eye(nx), goal=goal))
    task = Task(system)
    task.set_cost(cost)
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    task.set_init_obs(np.array([0.4, -0.2, 0.3, 0.1]))
    task.set_num_steps(T)
    ev = CandidateEvaluator(system, task, _hip_model(system, p))
    cands = [dict(horizon=8 + 2 * i, sigma=0.4, lmda=0.5, num_path=128, Q=np.full(nx, 1.0 + i),
                  R=np.full(nu, 0.05), F=np.ones(nx)) for i in range(4)]
    scores, obs, ctrls = ev.evaluate(cands, seed=3, return_trajectories=True)
    terms = cost_terms(cost, nx, nu)
    assert terms[0].tolist() == [1, 2, 0]
    for b in range(len(cands)):
        ref = score_terms(terms[0], terms[1], obs[b], ctrls[b])
        assert abs(scores[b] - ref) < 1e-10 * max(1.0, abs(ref))
    np.testing.assert_array_equal(scores, ev.evaluate(cands, seed=3))


def test_batch_tuner_with_true_dynamics_scores():
    """BatchPipelineTuner end to end on the device: surrogate scores from the batched closed
    loop, true-dynamics scores from MPPI.run() against a host callback."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import BatchPipelineTuner, CandidateEvaluator
    nx, nu, T = 3, 1, 12
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [64, 64], "tanh", seed=8)
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), np.eye(nx)))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    task.set_init_obs(np.array([0.5, -0.4, 0.3]))
    task.set_num_steps(T)
    ev = CandidateEvaluator(system, task, _hip_model(system, p))
    truth = MLPOracle(system, p)
    tuner = BatchPipelineTuner(system, ev, batch_size=4,
                               sampler=lambda n, rng: [dict(horizon=int(rng.integers(5, 12)),
                                                            sigma=float(rng.uniform(0.1, 1.0)),
                                                            lmda=float(rng.uniform(0.2, 1.5)),
                                                            num_path=128, Q=np.ones(nx), R=0.1 * np.ones(nu),
                                                            F=np.ones(nx)) for _ in range(n)])
    np.random.seed(0)
    best, res = tuner.run(6, np.random.default_rng(1), truedyn=lambda o, u: truth.pred(o, u))
    assert len(res.costs) == len(res.truedyn_costs) == 6
    assert np.all(np.isfinite(res.costs)) and np.all(np.isfinite(res.truedyn_costs))
    assert best is res.cfgs[int(np.argmin(res.costs))]
    # true dynamics == surrogate here, so both scores measure the same closed loop up to the
    # different noise streams: same order of magnitude
    ratio = np.array(res.truedyn_costs) / np.array(res.costs)
    assert np.all(ratio > 0.5) and np.all(ratio < 2.0)


# ---- the tuner's objective through eval_cfg's call shape (pipeline_tuner.py:213-258) -----------
def _evalcfg_stack(g):
    from autompc_amd import QuadCost, Task
    nx = int(g["nx"])
    system = make_system(nx, 1)
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], True)
    check_weights(p, g)
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    if "bounds" in g.files:
        task.set_ctrl_bounds([g["bounds"][0]], [g["bounds"][1]])
    task.set_init_obs(g["init"])
    task.set_num_steps(int(g["num_steps"]))
    if "term_thresh" in g.files:
        min_len, thresh = int(g["term_min_len"]), float(g["term_thresh"])
        task.set_term_cond(lambda traj: len(traj) >= min_len and abs(traj[-1].obs[0]) < thresh)
    return system, p, task


def _evalcfg_noise(g, n_ctl):
    """The draws eval_cfg's surrogate branch consumes from numpy's global stream: (H,1) at
    construction, (H,1) at reset(), then one (N,H,1) per control step."""
    N, H, scale = int(g["N"]), int(g["H"]), np.sqrt(float(g["sigma"]))
    np.random.seed(int(g["np_seed"]))
    np.random.normal(scale=scale, size=(H, 1))
    act0 = np.random.normal(scale=scale, size=(H, 1))
    eps = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(n_ctl)])
    return act0, eps


@pytest.mark.parametrize("name", ["loop_evalcfg_mppi", "loop_evalcfg_term"])
@pytest.mark.parametrize("check_every", [1, 4, 8])
def test_candidate_evaluator_scores_what_eval_cfg_scores(name, check_every):
    """CandidateEvaluator.evaluate() with no n_steps runs the episode eval_cfg simulates --
    task.term_cond (num_steps rows = num_steps - 1 controls, or the user's condition asked on the
    host between device segments), max_steps = num_steps -- and returns the reference's
    surr_cost, replaying the noise the reference consumed."""
    from autompc_amd.tuning import CandidateEvaluator
    g = golden(name)
    system, p, task = _evalcfg_stack(g)
    rows = len(g["surr_obs"])
    user_tc = "term_thresh" in g.files
    if not user_tc:
        assert rows == int(g["num_steps"])
    # with a termination condition the device runs up to check_every - 1 steps past the end; the
    # noise of those steps is whatever the stream holds next (it cannot influence the kept rows)
    n_noise = int(g["num_steps"]) if user_tc else rows - 1
    act0, eps = _evalcfg_noise(g, n_noise)
    ev = CandidateEvaluator(system, task, _hip_model(system, p), term_check_every=check_every)
    cand = dict(horizon=int(g["H"]), sigma=float(g["sigma"]), lmda=float(g["lmda"]),
                num_path=int(g["N"]), Q=g["Q"], R=g["R"], F=g["F"])
    scores, obs, ctrls = ev.evaluate([cand], eps_all=eps, act_init=act0, return_trajectories=True)
    assert ev.last_lengths.tolist() == [rows]
    assert obs.shape[1] == rows
    assert rel_err(obs[0], g["surr_obs"]) < 1e-9 and rel_err(ctrls[0], g["surr_ctrls"]) < 1e-9
    assert abs(scores[0] - g["surr_cost"]) < 1e-9 * abs(g["surr_cost"])


def test_term_cond_batch_is_per_candidate_and_batch_invariant():
    """Candidates that stop at different rows inside one batch (here: when the control energy spent
    so far exceeds a budget; one candidate never does and runs to max_steps): each is scored on its
    own rows, and its score is bit-identical to evaluating it alone with another segment length."""
    from autompc_amd.tuning import CandidateEvaluator
    g = golden("loop_evalcfg_term")
    system, p, task = _evalcfg_stack(g)
    budget, T = 0.6, int(g["num_steps"])
    task.set_term_cond(lambda traj: float(np.sum(traj.ctrls ** 2)) > budget)
    cands = [dict(horizon=6 + 2 * i, sigma=0.3 + 0.2 * i, lmda=0.5, num_path=64 + 32 * i,
                  Q=g["Q"] * (1.0 + i), R=g["R"] * rm, F=g["F"]) for i, rm in enumerate([1.0, 30.0, 1e3, 1e5])]
    ev = CandidateEvaluator(system, task, _hip_model(system, p), term_check_every=5)
    s, obs, ctrls = ev.evaluate(cands, seed=3, return_trajectories=True)
    lengths = ev.last_lengths.copy()
    print("rows per candidate:", lengths.tolist())
    assert len(set(lengths.tolist())) > 1 and lengths.max() <= T + 1
    for b, L in enumerate(lengths):
        # the condition first holds exactly at the candidate's last row (or max_steps was reached)
        energy = np.cumsum(np.sum(ctrls[b, :L] ** 2, axis=1))
        hits = [t + 1 for t in range(1, L) if energy[t - 1] > budget]      # asked with row t's control still zero
        assert (hits and hits[0] == L) or (not hits and L == T + 1)
        assert np.all(np.isnan(obs[b, L:])) and np.all(ctrls[b, L - 1] == 0.0)
        assert np.all(np.isfinite(obs[b, :L]))
    ev1 = CandidateEvaluator(system, task, _hip_model(system, p), term_check_every=1)
    for b in range(len(cands)):
        alone = ev1.evaluate([cands[b]], seed=3, index_offset=b)
        assert alone[0] == s[b] and ev1.last_lengths.tolist() == [lengths[b]]
    # a score is the task cost of exactly the rows kept
    cost = task.get_cost()
    from autompc_amd.trajectory import Trajectory
    for b, L in enumerate(lengths):
        ref = cost(Trajectory(system, int(L), obs[b, :L].copy(), ctrls[b, :L].copy()))
        assert abs(s[b] - ref) < 1e-10 * abs(ref)


def test_batch_tuner_columns_are_the_reference_eval_cfg_costs():
    """BatchPipelineTuner end to end against the reference's eval_cfg: the surrogate column (device
    closed loop) reproduces surr_cost, and the true-dynamics column (host simulate + device solves,
    numpy noise continuing the same global stream) reproduces truedyn_cost -- same episode length
    on both."""
    from autompc_amd.tuning import BatchPipelineTuner, CandidateEvaluator
    g = golden("loop_evalcfg_mppi")
    system, p, task = _evalcfg_stack(g)
    rows = len(g["surr_obs"])
    act0, eps = _evalcfg_noise(g, rows - 1)       # leaves the global stream where eval_cfg's
    truth = MLPOracle(system, p)                  # true-dynamics branch starts drawing
    cand = dict(horizon=int(g["H"]), sigma=float(g["sigma"]), lmda=float(g["lmda"]),
                num_path=int(g["N"]), Q=g["Q"], R=g["R"], F=g["F"])
    ev = CandidateEvaluator(system, task, _hip_model(system, p))
    tuner = BatchPipelineTuner(system, ev, batch_size=1, sampler=lambda n, rng: [cand] * n,
                               truedyn_noise="numpy", eval_kwargs=dict(eps_all=eps, act_init=act0))
    best, res = tuner.run(1, np.random.default_rng(0), truedyn=lambda o, u: truth.pred(o, u))
    assert abs(res.costs[0] - g["surr_cost"]) < 1e-9 * abs(g["surr_cost"])
    assert abs(res.truedyn_costs[0] - g["truedyn_cost"]) < 1e-9 * abs(g["truedyn_cost"])


def test_host_simulate_with_task_term_cond_matches_eval_cfg_ilqr():
    """eval_cfg's call shape for iLQR: reset(), simulate(controller, init_obs, task.term_cond,
    sim_model=surrogate, max_steps=num_steps) -> num_steps rows; every run() is a device solve."""
    from autompc_amd import IterativeLQR, simulate
    g = golden("loop_evalcfg_ilqr")
    system, p, task = _evalcfg_stack(g)
    model = _hip_model(system, p)
    ctl = IterativeLQR(system, task, model, int(g["H"]))
    ctl.reset()
    traj = simulate(ctl, task.get_init_obs(), task.term_cond, sim_model=model,
                    max_steps=task.get_num_steps())
    assert len(traj) == int(g["num_steps"])
    assert rel_err(traj.obs, g["surr_obs"]) < 1e-7 and rel_err(traj.ctrls, g["surr_ctrls"]) < 1e-7
    assert abs(task.get_cost()(traj) - g["surr_cost"]) < 1e-7 * abs(g["surr_cost"])


# ---- iLQR candidates: the tuner's other controller (IterativeLQRFactory, control/ilqr.py:31-41) --
def test_ilqr_candidate_evaluator_matches_eval_cfg_golden():
    """The reference's eval_cfg episode with an iLQR controller (loop_evalcfg_ilqr.npz) through the
    batched evaluator, and the tuner's two cost columns on top of it."""
    from autompc_amd.tuning import BatchPipelineTuner, IlqrCandidateEvaluator
    g = golden("loop_evalcfg_ilqr")
    system, p, task = _evalcfg_stack(g)
    ev = IlqrCandidateEvaluator(system, task, _hip_model(system, p))
    cand = dict(horizon=int(g["H"]), Q=g["Q"], R=g["R"], F=g["F"])
    scores, obs, ctrls = ev.evaluate([cand], return_trajectories=True)
    assert ev.last_lengths.tolist() == [int(g["num_steps"])]
    assert rel_err(obs[0], g["surr_obs"]) < 1e-7 and rel_err(ctrls[0], g["surr_ctrls"]) < 1e-7
    assert abs(scores[0] - g["surr_cost"]) < 1e-7 * abs(g["surr_cost"])
    truth = MLPOracle(system, p)
    tuner = BatchPipelineTuner(system, ev, batch_size=1, sampler=lambda n, rng: [cand] * n)
    best, res = tuner.run(1, np.random.default_rng(0), truedyn=lambda o, u: truth.pred(o, u))
    assert abs(res.costs[0] - g["surr_cost"]) < 1e-7 * abs(g["surr_cost"])
    assert abs(res.truedyn_costs[0] - g["truedyn_cost"]) < 1e-7 * abs(g["truedyn_cost"])


def test_ilqr_candidate_batch_equals_the_drop_in_controller_per_candidate():
    """Heterogeneous iLQR candidates (horizons 5-25, gains over several decades, bounded controls) in
    one batch: every candidate's closed loop is the one host simulate() + the drop-in IterativeLQR
    produce for it alone; a candidate whose Quu is singular scores inf like eval_cfg's LinAlgError
    branch (pipeline_tuner.py:236-239) without disturbing the others."""
    from autompc_amd import IterativeLQR, QuadCost, Task, simulate
    from autompc_amd.tuning import IlqrCandidateEvaluator, random_ilqr_candidates
    nx, nu, T = 4, 2, 9
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [64, 48], "tanh", seed=21)
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), 2.0 * np.eye(nx)))
    task.set_ctrl_bounds(-0.6 * np.ones(nu), 0.6 * np.ones(nu))
    task.set_init_obs(np.array([0.3, -0.2, 0.25, 0.1]))
    task.set_num_steps(T)
    model = _hip_model(system, p)
    cands = random_ilqr_candidates(system, 6, seed=4)
    for c in cands:
        c["Q"], c["R"], c["F"] = c["Q"] ** 0.3, c["R"] ** 0.3, c["F"] ** 0.3
    cands[1]["horizon"] = cands[0]["horizon"]                   # two candidates share a plan
    ev = IlqrCandidateEvaluator(system, task, model)
    scores, obs, ctrls = ev.evaluate(cands, return_trajectories=True)
    assert np.all(np.isfinite(scores))
    for b, c in enumerate(cands):
        t1 = Task(system)
        t1.set_cost(QuadCost(system, np.diag(c["Q"]), np.diag(c["R"]), np.diag(c["F"])))
        t1.set_ctrl_bounds(-0.6 * np.ones(nu), 0.6 * np.ones(nu))
        ctl = IterativeLQR(system, t1, model, c["horizon"])
        ctl.reset()
        traj = simulate(ctl, task.get_init_obs(), task.term_cond, sim_model=model, max_steps=T)
        assert len(traj) == T
        assert rel_err(obs[b], traj.obs) < 1e-9 and rel_err(ctrls[b], traj.ctrls) < 1e-9
        assert abs(scores[b] - task.get_cost()(traj)) < 1e-9 * abs(scores[b])
    np.testing.assert_array_equal(ev.evaluate(cands[2:4]), scores[2:4])      # batch-invariant
    # singular Quu: the control has no effect on the model and R = 0
    p2 = {k: ([w.copy() for w in v] if isinstance(v, list) else v) for k, v in p.items()}
    p2["weights"][0][:, nx:] = 0.0
    ev2 = IlqrCandidateEvaluator(system, task, _hip_model(system, p2))
    good = dict(horizon=8, Q=np.ones(nx), R=0.1 * np.ones(nu), F=np.ones(nx))
    sing = dict(horizon=8, Q=np.ones(nx), R=np.zeros(nu), F=np.ones(nx))
    s2 = ev2.evaluate([good, sing, good])
    assert np.isinf(s2[1]) and np.isfinite(s2[0]) and s2[0] == s2[2]
