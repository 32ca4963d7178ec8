"""Per-candidate controller models (ampc_mppi_plan_set_models / ampc_ilqr_plan_set_models; they run on the
shape-specialised kernels: registered shapes, or -- the two-model golden's 3-state network -- the shape's
run-time compiled plugin, which set_models waits for): eval_cfg builds
the controller with pipeline(cfg, task, trajs), which instantiates a model PER CONFIGURATION when the pipeline
has a model factory (pipeline.py:138-145, tuning/pipeline_tuner.py:213-231).  The reference's two-model golden
(tests/golden/loop_evalcfg_twomodels.npz) evaluated in ONE batch, and the property that a candidate's score
does not depend on which other models share the batch.  Needs MI355X."""
import numpy as np
import pytest

from conftest import golden
from helpers import golden_params, make_system, rel_err, weight_checksum
from oracle import mlp as omlp

pytestmark = pytest.mark.gpu


def _mlp(system, p):
    from autompc_amd import MLP
    m = MLP(system, n_hidden_layers=len(p["weights"]) - 1, nonlintype=p["activation"],
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    return m


def _stack(g):
    from autompc_amd import QuadCost, Task
    nx = int(g["nx"])
    system = make_system(nx, 1, dt=float(g["dt"]))
    pa = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed_a"], True)
    pb = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed_b"], True)
    np.testing.assert_allclose(weight_checksum(pa), g["wsum_a"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(weight_checksum(pb), g["wsum_b"], rtol=0, atol=1e-12)

    def task(T, bounded):
        t = Task(system)
        t.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
        if bounded:
            t.set_ctrl_bounds([g["bounds"][0]], [g["bounds"][1]])
        t.set_init_obs(g["init"])
        t.set_num_steps(T)
        return t
    return system, _mlp(system, pa), _mlp(system, pb), task


def _noise(g, tag, n_ctl):
    """The draws eval_cfg's surrogate branch consumes from numpy's global stream: (H,1) at construction,
    (H,1) at reset(), then one (N,H,1) per control step."""
    N, H, scale = int(g["N"]), int(g["H"]), np.sqrt(float(g["sigma"]))
    np.random.seed(int(g["mppi_%s_np_seed" % tag]))
    np.random.normal(scale=scale, size=(H, 1))
    act0 = np.random.normal(scale=scale, size=(H, 1))
    eps = np.stack([np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(n_ctl)])
    return act0, eps


@pytest.mark.parametrize("order", ["ab", "ba"])
def test_two_models_in_one_mppi_batch_reproduce_the_references_eval_cfg(order):
    from autompc_amd.tuning import CandidateEvaluator
    g = golden("loop_evalcfg_twomodels")
    system, ma, mb, task = _stack(g)
    T = int(g["num_steps_mppi"])
    models = {"a": ma, "b": mb}
    ev = CandidateEvaluator(system, task(T, True), ma)           # model A is the surrogate (and the default model)
    cands, acts, epss = [], [], []
    for tag in order:
        c = dict(horizon=int(g["H"]), sigma=float(g["sigma"]), lmda=float(g["lmda"]), num_path=int(g["N"]),
                 Q=g["Q"], R=g["R"], F=g["F"])
        if tag == "b" or order == "ba":
            c["model"] = models[tag]                             # ("ab": candidate A relies on the default)
        a0, e = _noise(g, tag, T - 1)
        cands.append(c); acts.append(a0.ravel()); epss.append(e.reshape(T - 1, -1))
    scores, obs, ctrls = ev.evaluate(cands, eps_all=np.concatenate(epss, axis=1), act_init=np.concatenate(acts),
                                     return_trajectories=True)
    for k, tag in enumerate(order):
        assert rel_err(obs[k], g["mppi_%s_obs" % tag]) < 1e-9 and rel_err(ctrls[k], g["mppi_%s_ctrls" % tag]) < 1e-9
        assert abs(scores[k] - g["mppi_%s_cost" % tag]) < 1e-9 * abs(g["mppi_%s_cost" % tag])


@pytest.mark.parametrize("device_resident", [True, False])
def test_two_models_in_one_ilqr_batch_reproduce_the_references_eval_cfg(device_resident):
    from autompc_amd.tuning import IlqrCandidateEvaluator
    g = golden("loop_evalcfg_twomodels")
    system, ma, mb, task = _stack(g)
    T = int(g["num_steps_ilqr"])
    ev = IlqrCandidateEvaluator(system, task(T, False), ma, device_resident=device_resident)
    cands = [dict(horizon=int(g["ilqr_b_H"]), Q=g["Q"], R=g["R"], F=g["F"], model=mb),
             dict(horizon=int(g["ilqr_a_H"]), Q=g["Q"], R=g["R"], F=g["F"]),
             dict(horizon=int(g["ilqr_b_H"]), Q=g["Q"], R=g["R"], F=g["F"], model=mb)]
    scores, obs, ctrls = ev.evaluate(cands, return_trajectories=True)
    for k, tag in enumerate("bab"):
        assert rel_err(obs[k], g["ilqr_%s_obs" % tag]) < 1e-6 and rel_err(ctrls[k], g["ilqr_%s_ctrls" % tag]) < 1e-6
        assert abs(scores[k] - g["ilqr_%s_cost" % tag]) < 1e-6 * abs(g["ilqr_%s_cost" % tag])
    np.testing.assert_array_equal(obs[0], obs[2])


def _hc_models(n, hidden=(256, 256), nx=17, nu=6):
    system = make_system(nx, nu)
    return system, [_mlp(system, omlp.random_params(nx, nu, list(hidden), "relu", seed=70 + k)) for k in range(n)]


def test_a_score_does_not_depend_on_the_models_sharing_the_batch():
    """HalfCheetah-shaped batch with 4 distinct 2x256 models: every candidate's score (MPPI and iLQR) is bit
    for bit what it gets alone in a batch with only its own model; mixed SHAPES are split into plans per
    shape and give the same scores again."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import CandidateEvaluator, IlqrCandidateEvaluator, random_candidates, random_ilqr_candidates
    system, models = _hc_models(4)
    nx, nu = 17, 6
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.01 * np.eye(nu), np.eye(nx)))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    task.set_init_obs(np.random.default_rng(0).uniform(-0.1, 0.1, size=nx))
    task.set_num_steps(6)
    cands = random_candidates(system, 10, seed=2)
    for k, c in enumerate(cands):
        c["model"] = models[k % 4]
    ev = CandidateEvaluator(system, task, models[0])
    full = ev.evaluate(cands, seed=4)
    for k in (1, 6, 9):
        alone = CandidateEvaluator(system, task, models[0]).evaluate([cands[k]], seed=4, index_offset=k)
        assert alone[0] == full[k]
        own = CandidateEvaluator(system, task, cands[k]["model"], surrogate=models[0])
        c2 = {kk: v for kk, v in cands[k].items() if kk != "model"}
        assert own.evaluate([c2], seed=4, index_offset=k)[0] == full[k]
    assert len(set(np.round(full, 9).tolist())) == len(full)
    # a second architecture in the same batch: split by shape, same scores for the first ten
    _, small = _hc_models(1, hidden=(128, 128))
    extra = random_candidates(system, 3, seed=5)
    for c in extra:
        c["model"] = small[0]
    mixed = ev.evaluate(cands[:5] + extra + cands[5:], seed=4,
                        index_offset=np.concatenate([np.arange(5), 100 + np.arange(3), 5 + np.arange(5)]))
    np.testing.assert_array_equal(np.concatenate([mixed[:5], mixed[8:]]), full)
    assert np.all(np.isfinite(mixed))
    # iLQR
    ic = random_ilqr_candidates(system, 8, seed=3)
    for k, c in enumerate(ic):
        c["Q"], c["R"], c["F"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25
        c["model"] = models[(k + 1) % 4]
    iev = IlqrCandidateEvaluator(system, task, models[0], max_slots=3)
    ifull = iev.evaluate(ic)
    for k in (0, 5):
        own = IlqrCandidateEvaluator(system, task, ic[k]["model"], surrogate=models[0])
        c2 = {kk: v for kk, v in ic[k].items() if kk != "model"}
        assert own.evaluate([c2])[0] == ifull[k]


def test_batch_tuner_searches_over_models():
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import BatchPipelineTuner, CandidateEvaluator
    system, models = _hc_models(3, hidden=(64, 64), nx=4, nu=1)        # (a registered shape: CartPole, csrc/shapes.hpp)
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(4), 0.01 * np.eye(1), np.eye(4)))
    task.set_ctrl_bounds(-np.ones(1), np.ones(1))
    task.set_init_obs(np.array([0.1, -0.1, 0.05, 0.0]))
    task.set_num_steps(5)
    tuner = BatchPipelineTuner(system, CandidateEvaluator(system, task, models[0]), batch_size=12, models=models)
    best, res = tuner.run(24, np.random.default_rng(1), seed=2)
    assert len(res.costs) == 24 and np.all(np.isfinite(res.costs))
    assert {c["model_index"] for c in res.cfgs} == {0, 1, 2} and best is res.cfgs[int(np.argmin(res.costs))]
    with pytest.raises(RuntimeError, match="shape"):
        from autompc_amd import _lib
        h0, h1 = _lib.Handle(0, "f64"), _lib.Handle(0, "f64")
        models[0].stage_into(h0)
        _hc_models(1, hidden=(64, 48), nx=4, nu=1)[1][0].stage_into(h1)
        h0.set_quad_costs(np.eye(4), np.eye(1), np.eye(4), np.zeros(4))
        h0.set_ctrl_bounds(-np.ones(1), np.ones(1))
        plan = _lib.MppiPlan(h0, [64], [5], [1.0], [1.0])
        plan.set_models([h1], [0])


def test_candidates_with_and_without_a_model_run_on_their_own_models():
    """a candidate WITHOUT a "model" entry runs on the evaluator's model -- also when other candidates of
    the batch carry models of the same or of another shape, and when every candidate that carries one has a shape
    the evaluator's model does not have."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import CandidateEvaluator, random_candidates
    system, models = _hc_models(3, hidden=(64, 64), nx=4, nu=1)
    _, other = _hc_models(2, hidden=(64, 48), nx=4, nu=1)
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(4), 0.01 * np.eye(1), np.eye(4)))
    task.set_ctrl_bounds(-np.ones(1), np.ones(1))
    task.set_init_obs(np.array([0.1, -0.1, 0.05, 0.0]))
    task.set_num_steps(6)
    ev = CandidateEvaluator(system, task, models[0])
    cands = random_candidates(system, 6, seed=3)
    for c in cands:
        c["Q"], c["R"], c["F"], c["num_path"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25, 128
    alone = {}
    for i, c in enumerate(cands):                         # every candidate alone on each model it will meet
        for tag, m in (("default", None), ("same", models[1]), ("other", other[0])):
            alone[i, tag] = ev.evaluate([dict(c, **({} if m is None else {"model": m}))], seed=5, index_offset=i)[0]
    mixes = [["same", "default", "same", "default", "default", "same"],          # one shape: explicit + default
             ["other", "default", "same", "other", "default", "other"],          # two shapes, the first group's
             ["other", "other", "other", "other", "other", "other"]]             # first candidate is NOT the default
    for mix in mixes:
        batch = [dict(c, **({} if t == "default" else {"model": models[1] if t == "same" else other[0]}))
                 for c, t in zip(cands, mix)]
        got = ev.evaluate(batch, seed=5)
        for i, t in enumerate(mix):
            assert got[i] == alone[i, t], (mix, i)


def test_a_model_table_never_meets_the_run_time_shape_kernels_silently():
    """the run-time-shape kernels take one model per plan.  A plan with a model table refuses indicator
    cost terms (they live in those kernels), and a geometry rebuild keeps the table on the specialised kernels."""
    from autompc_amd import _lib
    system, models = _hc_models(2, hidden=(64, 64), nx=4, nu=1)
    hs = [_lib.Handle(0, "f64") for _ in range(2)]
    for h, m in zip(hs, models):
        m.stage_into(h)
    h0 = hs[0]
    h0.set_quad_costs(np.eye(4), 0.01 * np.eye(1), np.eye(4), np.zeros(4))
    h0.set_ctrl_bounds(-np.ones(1), np.ones(1))
    plan = _lib.MppiPlan(h0, [64, 64], [8, 8], [1.0, 1.0], [1.0, 1.0])
    plan.set_geometry(32, 30)
    plan.set_models(hs, [0, 1])
    x0 = np.tile([0.1, -0.1, 0.05, 0.0], (2, 1))
    act = np.zeros((2, 8, 1))
    plan.upload(x0, act, None)
    plan.generate_eps(1, 0)
    plan.solve()
    a1 = plan.download(act_seq=True, u=False)[0].reshape(2, 8).copy()
    plan.set_geometry(16, 30)                         # rebuild with the table in place: still per-problem models
    plan.upload(x0, act, None)
    plan.generate_eps(1, 0)
    plan.solve()
    a2 = plan.download(act_seq=True, u=False)[0].reshape(2, 8)
    assert rel_err(a2, a1) < 1e-9 and not np.array_equal(a1[0], a1[1])
    from autompc_amd.costs.terms import THRESHOLD
    thr = (np.array([THRESHOLD], dtype=np.int32), np.concatenate([np.zeros(4), [0.0, 4.0, 0.5]]))   # goal | obs range | threshold
    h0.set_indicator_costs(thr)
    with pytest.raises(_lib.AmpcError, match="table of controller models"):
        plan.solve()
    h0.set_indicator_costs(None)
    plan.solve()


def test_one_evaluation_models_never_build_kernels_and_share_no_table():
    """Models a tuner fits for ONE evaluation (jit_kernels = False; BatchPipelineTuner.fit_models sets it): handles
    that hold them never start -- or wait for -- the run-time build of shape-specialised kernels
    (ampc_handle_set_jit), and two such models that share an unregistered shape, which cannot share a plan through a
    model table, get a plan each on the run-time-shape kernels.  Scores are those of each candidate alone."""
    import time
    from autompc_amd import QuadCost, Task, _lib
    from autompc_amd.tuning import CandidateEvaluator, IlqrCandidateEvaluator, random_candidates, random_ilqr_candidates
    # (a shape nothing else in the suite uses: not registered, no plugin loaded or cached for it)
    system, models = _hc_models(3, hidden=(176, 24), nx=5, nu=2)
    for m in models[1:]:
        m.jit_kernels = False
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(5), 0.01 * np.eye(2), np.eye(5)))
    task.set_ctrl_bounds(-np.ones(2), np.ones(2))
    task.set_init_obs(np.array([0.1, -0.1, 0.05, 0.0, 0.02]))
    task.set_num_steps(6)
    h = _lib.Handle(0, "f64", jit=False)
    models[1].stage_into(h)
    h.set_quad_costs(np.eye(5), 0.01 * np.eye(2), np.eye(5), np.zeros(5))
    assert h.jit_status()[0] == 0                                         # no build was started
    h.close()
    cands = random_candidates(system, 4, seed=2)
    for c in cands:
        c["Q"], c["R"], c["F"], c["num_path"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25, 128
    batch = [dict(c, model=models[1 + i % 2]) for i, c in enumerate(cands)]
    ev = CandidateEvaluator(system, task, models[1])
    t0 = time.perf_counter()
    got = ev.evaluate(batch, seed=3)
    assert time.perf_counter() - t0 < 3.0                                # (a build takes 3-7 s: none was waited for)
    for i, c in enumerate(batch):
        np.testing.assert_array_equal(ev.evaluate([c], seed=3, index_offset=i), got[i:i + 1])
    ic = random_ilqr_candidates(system, 4, seed=2)
    for c in ic:
        c["Q"], c["R"], c["F"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25
    ibatch = [dict(c, model=models[1 + i % 2]) for i, c in enumerate(ic)]
    iev = IlqrCandidateEvaluator(system, task, models[1])
    igot = iev.evaluate(ibatch)
    for i, c in enumerate(ibatch):
        np.testing.assert_array_equal(iev.evaluate([c], index_offset=i), igot[i:i + 1])
    h2 = _lib.Handle(0, "f64", jit=False)
    models[2].stage_into(h2)
    h2.set_quad_costs(np.eye(5), 0.01 * np.eye(2), np.eye(5), np.zeros(5))
    assert h2.jit_status()[0] == 0                                        # ... and none by the evaluations either
    h2.close()
