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


def _nan_rows():
    """Rows with NaN entries: inside / outside the threshold term's range, next to violating and harmless entries."""
    g = golden("cost_terms")
    lo, hi = [int(v) for v in g["thr_range"]]
    base = g["goal"].copy()
    rows = []
    for nan_at, bump_at in ((lo, lo + 1), (lo, None), (hi if hi < 5 else 0, lo), (hi if hi < 5 else 0, None)):
        x = base.copy()
        if bump_at is not None:
            x[bump_at] += 10.0 * float(g["thr"])            # a violating entry
        x[nan_at] = np.nan
        rows.append(x)
    return g, np.array(rows)


def test_nan_observations_are_charged_as_the_reference_charges_them():
    """ADVICE r5.  ThresholdCost is norm(diff, inf) > thr (thresh_cost.py:27-32): one NaN inside the range makes the
    norm NaN and the term is NOT charged, whatever the other entries do; a NaN outside the range is invisible.
    BoxThresholdCost compares entry by entry (:73-77): a NaN entry violates nothing, the others still count.
    An empty obs_range raises in the reference (maximum of an empty slice): cost_terms refuses it."""
    g, rows = _nan_rows()
    system, costs = _costs(g)
    lo, hi = [int(v) for v in g["thr_range"]]
    for x in rows:
        in_range_nan = bool(np.isnan(x[lo:hi]).any())
        finite = np.nan_to_num(x, nan=float(g["goal"][0]))
        viol = bool(np.max(np.abs(finite[lo:hi] - g["goal"][lo:hi])) > float(g["thr"]))
        want = 0.0 if in_range_nan else float(viol)
        with np.errstate(invalid="ignore"):
            assert costs["thresh"].eval_obs_cost(x) == want
            kinds, params = cost_terms(costs["thresh"], 5, 3)
            assert score_terms(kinds, params, x[None, :], np.zeros((1, 3))) == want
            lim = g["limits"]
            box_want = float(bool((x < lim[:, 0]).any() or (x > lim[:, 1]).any()))
            assert costs["box"].eval_obs_cost(x) == box_want
    with pytest.raises(ValueError, match="empty obs_range"):
        cost_terms(ThresholdCost(system, g["goal"], [2, 2], 0.5), 5, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_device_scores_of_nan_rows(precision):
    from autompc_amd import _lib
    g, rows = _nan_rows()
    _, costs = _costs(g)
    obs = np.repeat(rows[:, None, :], 3, axis=1)               # [B, T, no]: three identical rows each
    ctl = np.zeros((rows.shape[0], 3, 3))
    h = _lib.Handle(0, precision)
    for name in ("thresh", "box", "sum_tb"):
        terms = cost_terms(costs[name], 5, 3)
        got = h.score_trajectories(terms, obs, ctl)
        with np.errstate(invalid="ignore"):
            want = np.array([score_terms(terms[0], terms[1], o, c) for o, c in zip(obs, ctl)])
        np.testing.assert_array_equal(got, want)
    h.close()
