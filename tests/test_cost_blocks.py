"""Host logic of the affine-quadratic cost block (autompc_amd/costs/blocks.py): a sum of quadratic
terms with different goals as ONE device block.  CPU only; the values are checked against the
reference's own SumCost outputs (tests/golden/cost_sum.npz)."""
import numpy as np
import pytest

from conftest import golden
from helpers import make_system
from autompc_amd import QuadCost, ThresholdCost
from autompc_amd.costs import SumCost
from autompc_amd.costs.blocks import is_quad_sum, quad_sum_block


def _sum_from(g, kind, system, **kw):
    ts = [QuadCost(system, q, r, f, goal=gl, **kw)
          for q, r, f, gl in zip(g[kind + "_Qs"], g[kind + "_Rs"], g[kind + "_Fs"], g[kind + "_goals"])]
    return SumCost(system, ts)


def _stage(b, x):
    d = x - b["goal"]
    return d @ b["Q"] @ d + b["lin"] @ d + b["consts"][0]


def _term(b, x):
    d = x - b["goal"]
    return d @ b["F"] @ d + b["lin_term"] @ d + b["consts"][1]


@pytest.mark.parametrize("kind", ["gauss", "dense", "three", "samegoal"])
def test_block_reproduces_the_references_sum(kind):
    g = golden("cost_sum")
    system = make_system(5, 3)
    cost = _sum_from(g, kind, system)
    b = quad_sum_block(cost, 5, 3)
    obs, ctrl = g[kind + "_obs"], g[kind + "_ctrl"]
    assert abs(_stage(b, obs) - g[kind + "_obs_cost"]) < 1e-12 * max(1.0, abs(g[kind + "_obs_cost"]))
    assert abs(_term(b, obs) - g[kind + "_term_cost"]) < 1e-12 * max(1.0, abs(g[kind + "_term_cost"]))
    assert abs(ctrl @ b["R"] @ ctrl - g[kind + "_ctrl_cost"]) < 1e-12
    # gradient / Hessian of the stage cost as iLQR's sweep forms them from the block
    np.testing.assert_allclose((b["Q"] + b["Q"].T) @ (obs - b["goal"]) + b["lin"], g[kind + "_obs_j"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b["Q"] + b["Q"].T, g[kind + "_obs_h"], rtol=1e-13, atol=1e-13)
    # terminal gradient: the reference's ignores every term's goal (cost.py:195) -> (F+F') x, no affine part
    np.testing.assert_allclose((b["F"] + b["F"].T) @ obs, g[kind + "_term_j"], rtol=1e-12, atol=1e-12)
    assert b["terminal_goal"] is False
    # the product's own SumCost evaluates like the reference's
    assert abs(cost.eval_obs_cost(obs) - g[kind + "_obs_cost"]) < 1e-12 * max(1.0, abs(g[kind + "_obs_cost"]))
    if kind == "samegoal":     # shared goal: the affine part is EXACTLY zero -> the plain quadratic block
        assert not b["lin"].any() and not b["lin_term"].any() and not b["consts"].any()


def test_nonstrict_terms_put_the_goal_into_the_terminal_gradient():
    g = golden("cost_sum")
    system = make_system(5, 3)
    cost = _sum_from(g, "dense", system, strict_reference=False)
    b = quad_sum_block(cost, 5, 3)
    assert b["terminal_goal"] is True
    x = g["dense_obs"]
    want = sum((f + f.T) @ (x - gl) for f, gl in zip(g["dense_Fs"], g["dense_goals"]))
    np.testing.assert_allclose((b["F"] + b["F"].T) @ (x - b["goal"]) + b["lin_term"], want, rtol=1e-12, atol=1e-12)
    mixed = SumCost(system, [cost.costs[0], QuadCost(system, np.eye(5), np.eye(3), goal=np.ones(5))])
    with pytest.raises(TypeError):
        quad_sum_block(mixed, 5, 3)
    # ... and the controllers' is_compatible (is_quad_sum) does not promise what quad_sum_block refuses
    assert is_quad_sum(cost) and not is_quad_sum(mixed)


class _RefLikeQuad:
    """Duck type of the reference's QuadCost (cost.py:43-64)."""
    is_quad = True

    def __init__(self, Q, R, F, goal):
        self._Q, self._R, self._F, self._goal = Q, R, F, goal

    def get_cost_matrices(self):
        return np.copy(self._Q), np.copy(self._R), np.copy(self._F)

    def get_goal(self):
        return np.copy(self._goal)


class _RefLikeSum:
    """Duck type of the reference's SumCost: ``get_goal`` returns the first COST OBJECT
    (sum_cost.py:45-47) and ``is_quad`` is False as soon as two goals differ (:84-93)."""

    def __init__(self, costs):
        self._costs = costs

    @property
    def costs(self):
        return self._costs[:]

    def get_goal(self):
        return self.costs[0]

    @property
    def is_quad(self):
        g = self.costs[0].get_goal()
        return all((g == c.get_goal()).all() for c in self.costs[1:])


def test_reference_style_sums_are_read_structurally():
    rng = np.random.default_rng(0)
    mk = lambda goal: _RefLikeQuad(rng.normal(size=(3, 3)), np.eye(2), rng.normal(size=(3, 3)), goal)   # noqa: E731
    g1 = rng.normal(size=3)
    same = _RefLikeSum([mk(g1), mk(g1)])
    assert same.is_quad and not isinstance(same.get_goal(), np.ndarray)     # what crashed in round 3
    b = quad_sum_block(same, 3, 2)
    np.testing.assert_array_equal(b["goal"], g1)
    assert not b["lin"].any()
    diff = _RefLikeSum([mk(g1), _RefLikeSum([mk(-g1), mk(2 * g1)])])         # nested, goals differ
    assert not diff.is_quad and is_quad_sum(diff)
    b = quad_sum_block(diff, 3, 2)
    x = rng.normal(size=3)
    leaves = [diff.costs[0]] + diff.costs[1].costs
    want = sum((x - c._goal) @ c._Q @ (x - c._goal) for c in leaves)
    assert abs(_stage(b, x) - want) < 1e-12 * abs(want)


def test_terms_without_a_quadratic_form_are_refused():
    system = make_system(3, 1)
    q = QuadCost(system, np.eye(3), np.eye(1))
    t = ThresholdCost(system, np.zeros(3), [0, 3], 0.5)
    assert is_quad_sum(q) and not is_quad_sum(q + t)
    with pytest.raises(TypeError):
        quad_sum_block(q + t, 3, 1)


def test_mppi_cost_parts_split_quadratic_and_indicator_terms():
    """MPPI takes sums of quadratic, threshold and box terms (mppi.py:73-82 charges any Cost term by term);
    iLQR's is_quad_sum stays quadratic-only."""
    from autompc_amd import BoxThresholdCost
    from autompc_amd.costs.blocks import is_mppi_cost, mppi_cost_parts
    g = golden("cost_sum")
    system = make_system(5, 3)
    quad = _sum_from(g, "dense", system)
    goal = np.arange(5) * 0.1
    thr = ThresholdCost(system, goal, [1, 4], 0.3)
    limits = np.array([[-1.0, 1.0], [-np.inf, 0.5], [-0.2, np.inf], [-np.inf, np.inf], [-2.0, 2.0]])
    box = BoxThresholdCost(system, limits)
    cost = SumCost(system, [quad, thr, box])
    assert is_mppi_cost(cost) and is_mppi_cost(box) and is_mppi_cost(quad) and not is_quad_sum(cost)
    blk, terms = mppi_cost_parts(cost, 5, 3)
    ref = quad_sum_block(quad, 5, 3)
    for k in ("Q", "R", "F", "goal", "lin", "lin_term", "consts"):
        np.testing.assert_array_equal(blk[k], ref[k])
    kinds, params = terms
    assert kinds.tolist() == [1, 2]
    np.testing.assert_array_equal(params, np.concatenate([goal, [1, 4, 0.3], limits[:, 0], limits[:, 1]]))
    blk0, terms0 = mppi_cost_parts(box, 5, 3)
    assert not any(np.any(blk0[k]) for k in ("Q", "R", "F", "lin", "lin_term", "consts")) and terms0[0].tolist() == [2]
    assert mppi_cost_parts(quad, 5, 3)[1] is None

    class Other:
        is_quad = False
    assert not is_mppi_cost(SumCost(system, [quad, Other()]))
    with pytest.raises(TypeError):
        mppi_cost_parts(SumCost(system, [quad, Other()]), 5, 3)
    assert not is_mppi_cost(SumCost(system, [quad] + [thr] * 9))          # at most 8 indicator terms
