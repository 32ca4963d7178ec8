"""Trajectory scoring with threshold / box / summed costs.

CPU: the oracle's score_terms and the host cost classes + cost_terms flattening against vectors
made by the reference's own ThresholdCost / BoxThresholdCost / SumCost (gen_golden.py
gen_cost_terms).  GPU: ampc_score_trajectories against the same vectors."""
import numpy as np
import pytest

from conftest import golden
from helpers import make_system, rel_err
from autompc_amd import zeros
from autompc_amd.costs import BoxThresholdCost, QuadCost, ThresholdCost, cost_terms
from oracle.costs import score_terms


def _costs(g):
    system = make_system(5, 3)
    quad = QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"])
    quad2 = QuadCost(system, np.eye(5), np.eye(3), 2 * np.eye(5), goal=-g["goal"])
    thresh = ThresholdCost(system, g["goal"], [int(v) for v in g["thr_range"]], float(g["thr"]))
    box = BoxThresholdCost(system, g["limits"], goal=g["goal"])
    return system, {"quad": quad, "thresh": thresh, "box": box, "sum_tb": thresh + box,
                    "sum_all": quad + thresh + box + quad2}


CASES = ["quad", "thresh", "box", "sum_tb", "sum_all"]


@pytest.mark.parametrize("name", CASES)
def test_host_costs_and_oracle_match_reference(name):
    g = golden("cost_terms")
    system, costs = _costs(g)
    c = costs[name]
    kinds, params = cost_terms(c, 5, 3)
    for b in range(g["obs"].shape[0]):
        traj = zeros(system, g["obs"].shape[1])
        traj.obs[:] = g["obs"][b]
        traj.ctrls[:] = g["ctrls"][b]
        want = g["score_" + name][b]
        assert abs(c(traj) - want) <= 1e-12 * max(1.0, abs(want))
        got = score_terms(kinds, params, g["obs"][b], g["ctrls"][b])
        assert abs(got - want) <= 1e-12 * max(1.0, abs(want))


def test_flattening_layout_and_errors():
    g = golden("cost_terms")
    system, costs = _costs(g)
    kinds, params = cost_terms(costs["sum_all"], 5, 3)
    assert kinds.tolist() == [0, 1, 2, 0]
    assert params.size == 2 * (2 * 25 + 9 + 5) + 8 + 10

    class Odd:
        is_quad = False
    with pytest.raises(TypeError):
        cost_terms(Odd(), 5, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("name", CASES)
def test_device_scores_match_reference(name, precision):
    from autompc_amd import _lib
    g = golden("cost_terms")
    _, costs = _costs(g)
    h = _lib.Handle(0, precision)
    got = h.score_trajectories(cost_terms(costs[name], 5, 3), g["obs"], g["ctrls"])
    h.close()
    want = g["score_" + name]
    if name in ("thresh", "box", "sum_tb"):
        # indicator counts are exact in both precisions (the rows placed exactly on a limit
        # are representable in f32 only by luck, so f32 is checked away from those rows)
        if precision == "f64":
            np.testing.assert_array_equal(got, want)
        else:
            assert np.max(np.abs(got - want)) <= 1.0
    else:
        assert rel_err(got, want) < (1e-12 if precision == "f64" else 2e-5)


@pytest.mark.gpu
def test_device_scorer_rejects_bad_terms():
    from autompc_amd import _lib
    h = _lib.Handle(0, "f64")
    obs, ctl = np.zeros((2, 3, 4)), np.zeros((2, 3, 1))
    with pytest.raises(_lib.AmpcError):
        h.score_trajectories((np.array([7], dtype=np.int32), np.zeros(4)), obs, ctl)
    with pytest.raises(_lib.AmpcError):   # threshold range outside the observation
        h.score_trajectories((np.array([1], dtype=np.int32),
                              np.concatenate([np.zeros(4), [0, 9, 0.5]])), obs, ctl)
    h.close()


def test_reference_style_cost_objects_flatten_too():
    """cost_terms recognises terms structurally, so the reference's own objects (attributes
    _obs_range / _threshold / _limits, a `costs` property on sums) work unchanged."""
    class RefThreshold:                       # attribute layout of autompc.costs.ThresholdCost
        def __init__(self, goal, obs_range, threshold):
            self._goal, self._obs_range, self._threshold = np.array(goal), list(obs_range), threshold
            self.is_quad = False

    class RefBox:                             # ... of autompc.costs.BoxThresholdCost
        def __init__(self, limits):
            self._limits = np.array(limits)
            self.is_quad = False

    class RefSum:                             # ... of autompc.costs.SumCost
        def __init__(self, costs):
            self._costs = costs

        @property
        def costs(self):
            return self._costs[:]

    g = golden("cost_terms")
    ref = RefSum([RefThreshold(g["goal"], g["thr_range"], float(g["thr"])), RefBox(g["limits"])])
    kinds, params = cost_terms(ref, 5, 3)
    _, own = _costs(g)
    k2, p2 = cost_terms(own["sum_tb"], 5, 3)
    np.testing.assert_array_equal(kinds, k2)
    np.testing.assert_array_equal(params, p2)
    for b in range(g["obs"].shape[0]):
        assert score_terms(kinds, params, g["obs"][b], g["ctrls"][b]) == g["score_sum_tb"][b]
