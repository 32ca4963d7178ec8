"""The REFERENCE's own objects through this package's host layer.

Build-container only: needs /root/reference (the real williamedwards/autompc, imported unmodified the
way tests/golden/gen_golden.py imports it); skipped wherever the reference is absent (the GPU box).
Nothing of the reference is stored: the checks are made live against its classes --
``QuadCost`` / ``SumCost`` / ``ThresholdCost`` / ``BoxThresholdCost`` (autompc/costs/), the cost
factories, ``Task`` (tasks/task.py), ``Trajectory`` (trajectory.py), ``System`` -- pushed through
``quad_sum_block`` / ``cost_terms`` / ``episode_of`` / ``traj_to_state`` and the controllers'
constructors, i.e. everything INTEGRATION.md claims accepts "reference objects as they are".
"""
import os
import sys

import numpy as np
import pytest

if not os.path.isdir("/root/reference/autompc"):
    pytest.skip("needs the reference tree (/root/reference): build container only", allow_module_level=True)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gen_golden as gg                                              # noqa: E402  (imports the reference)
from autompc.costs import BoxThresholdCost, QuadCost, SumCost, ThresholdCost   # noqa: E402
from autompc.tasks import Task                                       # noqa: E402

from autompc_amd.costs.blocks import is_quad_sum, quad_sum_block     # noqa: E402
from autompc_amd.costs.terms import cost_terms                       # noqa: E402
from oracle.costs import score_terms                                 # noqa: E402


def _ref_traj(system, obs, ctrls):
    traj = gg.ampc.zeros(system, len(obs))
    traj.obs[:] = obs
    traj.ctrls[:] = ctrls
    return traj


@pytest.mark.parametrize("kind", ["gauss", "dense", "three", "samegoal"])
def test_reference_sum_costs_stage_as_one_block(kind):
    """The reference's SumCost -- incl. the factories' QuadCostFactory + GaussRegFactory product and
    the same-goal sum whose get_goal() returns a cost object (sum_cost.py:45-47) -- becomes a device
    block whose value / gradient / Hessian are the ones the reference's eval_* return."""
    system = gg.make_system(5, 3)
    cost = gg.sum_cost_of(system, kind, 123)
    assert isinstance(cost, SumCost) and is_quad_sum(cost)
    if kind == "samegoal":
        assert cost.is_quad and not isinstance(cost.get_goal(), np.ndarray)    # the round-3 crash
    else:
        assert not cost.is_quad
    b = quad_sum_block(cost, 5, 3)
    rng = np.random.default_rng(1)
    for _ in range(5):
        x, u = rng.normal(size=5), rng.normal(size=3)
        d = x - b["goal"]
        c, j, h = cost.eval_obs_cost_hess(x)
        assert abs(d @ b["Q"] @ d + b["lin"] @ d + b["consts"][0] - c) < 1e-11 * max(1.0, abs(c))
        np.testing.assert_allclose((b["Q"] + b["Q"].T) @ d + b["lin"], j, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(b["Q"] + b["Q"].T, h, rtol=1e-13, atol=1e-13)
        c, j, h = cost.eval_ctrl_cost_hess(u)
        assert abs(u @ b["R"] @ u - c) < 1e-12 * max(1.0, abs(c))
        np.testing.assert_allclose((b["R"] + b["R"].T) @ u, j, rtol=1e-12, atol=1e-12)
        ct = cost.eval_term_obs_cost(x)
        assert abs(d @ b["F"] @ d + b["lin_term"] @ d + b["consts"][1] - ct) < 1e-11 * max(1.0, abs(ct))
        _, tj, th = cost.eval_term_obs_cost_hess(x)                  # goal-less, term by term (cost.py:195)
        np.testing.assert_allclose((b["F"] + b["F"].T) @ x, tj, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(b["F"] + b["F"].T, th, rtol=1e-13, atol=1e-13)
    assert b["terminal_goal"] is False


def test_reference_cost_objects_flatten_to_score_terms():
    """ThresholdCost / BoxThresholdCost / QuadCost / their SumCost -> cost_terms -> the reference's
    own Cost.__call__ (cost.py:27-41) on random trajectories (device scorer's input format; the
    scorer itself is pinned against the same numbers in tests/test_cost_terms.py)."""
    system = gg.make_system(4, 2)
    rng = np.random.default_rng(5)
    goal = rng.normal(scale=0.3, size=4)
    quad = QuadCost(system, rng.normal(size=(4, 4)), np.diag(rng.uniform(0.1, 1, size=2)), np.eye(4), goal=goal)
    thr = ThresholdCost(system, goal, [1, 3], 0.4)
    box = BoxThresholdCost(system, np.array([[-1.0, 1.0], [-np.inf, 0.5], [-0.5, np.inf], [-np.inf, np.inf]]))
    for cost in (quad, thr, box, thr + box, quad + thr + box + quad):
        kinds, params = cost_terms(cost, 4, 2)
        for _ in range(3):
            obs, ctrls = rng.normal(scale=0.7, size=(8, 4)), rng.normal(size=(8, 2))
            want = cost(_ref_traj(system, obs, ctrls))
            assert abs(score_terms(kinds, params, obs, ctrls) - want) < 1e-11 * max(1.0, abs(want))
    with pytest.raises(TypeError):
        quad_sum_block(quad + thr, 4, 2)           # an indicator term has no quadratic block


def test_reference_task_and_trajectory_through_the_plugins():
    """Controllers and evaluators constructed on the reference's System / Task / costs; its
    Trajectory through traj_to_state; its Task through episode_of."""
    from autompc_amd import MLP, MPPI, IterativeLQR
    from autompc_amd.tuning.batch_eval import (CandidateEvaluator, IlqrCandidateEvaluator, candidate_cost_blocks,
                                               default_episode_controls, episode_of, score_trajectories)
    system = gg.make_system(3, 1)
    cost = gg.sum_cost_of(system, "gauss", 77)
    task = Task(system)
    task.set_cost(cost)
    task.set_ctrl_bound("u0", -1.0, 1.0)
    task.set_init_obs(np.array([0.1, 0.2, -0.1]))
    task.set_num_steps(12)
    model = MLP(system, n_hidden_layers=2, hidden_size_1=32, hidden_size_2=32, nonlintype="tanh")
    assert MPPI.is_compatible(system, task, model) and IterativeLQR.is_compatible(system, task, model)
    np.random.seed(0)
    mppi = MPPI(system, task, model, horizon=8, num_path=64)
    ilqr = IterativeLQR(system, task, model, 10)
    traj = _ref_traj(system, np.arange(12.0).reshape(4, 3), np.array([[0.1], [0.2], [0.3], [0.4]]))
    np.testing.assert_array_equal(mppi.traj_to_state(traj), [9, 10, 11, 0.4])        # mppi.py:170-173
    np.testing.assert_array_equal(ilqr.traj_to_state(traj), [9, 10, 11])             # ilqr.py:96-98
    np.testing.assert_array_equal(IterativeLQR(system, task, model, 10, strict_reference=False).traj_to_state(traj),
                                  [9, 10, 11, 0.4])
    # episode semantics read off the reference's Task (tasks/task.py:41-53,92-101)
    assert episode_of(task) == (12, None) and default_episode_controls(task) == 11
    task.set_term_cond(lambda tr: len(tr) >= 5)
    n, tc = episode_of(task)
    assert n == 12 and tc is not None and tc(traj) is False
    task.set_num_steps(12)
    # evaluators accept the task; a candidate may carry the reference's cost object
    ev = CandidateEvaluator(system, task, model)
    np.testing.assert_array_equal(ev.goal, np.asarray(cost.costs[0].get_goal()))
    IlqrCandidateEvaluator(system, task, model)
    blocks, tg = candidate_cost_blocks([dict(cost=cost), dict(Q=np.ones(3), R=np.ones(1), F=np.ones(3))], ev.goal, 3, 1)
    assert blocks["Q"].shape == (2, 3, 3) and blocks["lin"][0].any() and not blocks["lin"][1].any() and tg is False
    # host-side scoring of trajectories under the reference's sum cost == its own __call__
    rng = np.random.default_rng(2)
    obs, ctrls = rng.normal(size=(3, 6, 3)), rng.normal(size=(3, 6, 1))
    want = [cost(_ref_traj(system, obs[b], ctrls[b])) for b in range(3)]
    np.testing.assert_allclose(score_trajectories(cost, obs, ctrls), want, rtol=1e-11)


def test_reference_threshold_costs_on_nan_rows_and_empty_ranges():
    """What the REFERENCE's ThresholdCost / BoxThresholdCost do with NaN observations and an empty obs_range
    (thresh_cost.py:27-32, 73-77) is what the host classes, the flattened terms and the oracle do."""
    from autompc_amd.costs import ThresholdCost as OurThreshold
    from helpers import make_system
    system = gg.make_system(5, 3)
    goal = np.array([0.1, -0.2, 0.3, 0.0, 0.5])
    thr = ThresholdCost(system, goal, [1, 4], 0.4)
    box = BoxThresholdCost(system, np.array([[-1.0, 1.0], [-np.inf, 0.5], [-0.5, np.inf], [-1, 1], [-np.inf, np.inf]]))
    kinds, params = cost_terms(thr + box, 5, 3)
    rows = []
    for nan_at, bump_at in ((1, 2), (2, None), (0, 3), (4, None), (3, 0)):
        x = goal.copy()
        if bump_at is not None:
            x[bump_at] += 5.0
        x[nan_at] = np.nan
        rows.append(x)
    with np.errstate(invalid="ignore"):
        for x in rows:
            want = thr.eval_obs_cost(x) + box.eval_obs_cost(x)
            assert score_terms(kinds, params, x[None, :], np.zeros((1, 3))) == want
    empty = ThresholdCost(system, goal, [2, 2], 0.4)
    with pytest.raises(ValueError):
        empty.eval_obs_cost(goal)                                   # numpy: maximum of an empty slice
    with pytest.raises(ValueError):
        cost_terms(empty, 5, 3)
    with pytest.raises(ValueError):
        cost_terms(OurThreshold(make_system(5, 3), goal, [2, 2], 0.4), 5, 3)
