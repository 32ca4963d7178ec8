"""Shared builders for parity tests: regenerate the exact inputs the golden
vectors were produced from (tests/golden/gen_golden.py uses the same recipe)."""
import numpy as np

from autompc_amd import System
from oracle import mlp as omlp
from oracle.costs import BoxCostOracle, QuadCostOracle, SumCostOracle, ThresholdCostOracle


def make_system(nx, nu, dt=0.05):
    return System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=dt)


def normalisers(nx, nu, seed):
    rng = np.random.default_rng(seed + 7919)
    return (rng.normal(scale=0.3, size=nx + nu), rng.uniform(0.5, 2.0, size=nx + nu),
            rng.normal(scale=0.02, size=nx), rng.uniform(0.05, 0.2, size=nx))


def golden_params(nx, nu, hidden, activation, seed, plain_norm=False):
    p = omlp.random_params(nx, nu, [int(h) for h in hidden], str(activation), seed=int(seed))
    if not plain_norm:
        p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"] = normalisers(nx, nu, int(seed))
    return p


def weight_checksum(p):
    return np.array([sum(float(np.sum(w)) for w in p["weights"]),
                     sum(float(np.sum(np.abs(b))) for b in p["biases"]),
                     float(p["weights"][0][0, 0]), float(p["weights"][-1][-1, -1])])


def check_weights(p, g):
    np.testing.assert_allclose(weight_checksum(p), g["wsum"], rtol=0, atol=1e-12,
                               err_msg="numpy RNG drifted: golden MLP weights not reproducible")


def cost_from_golden(g):
    """The fixture's controller cost for the oracle: one QuadCost, or -- fixtures that store per-term
    arrays Qs / Rs / Fs / goals (gen_golden.gen_sumcost) -- the sum of quadratic terms."""
    if "Qs" in g:
        return SumCostOracle.from_arrays(g["Qs"], g["Rs"], g["Fs"], g["goals"])
    return QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"])


def hip_cost_from_golden(system, g):
    """The same cost as product objects (autompc_amd.QuadCost, summed with ``+``)."""
    from autompc_amd import QuadCost
    if "Qs" in g:
        terms = [QuadCost(system, q, r, f, goal=gl) for q, r, f, gl in zip(g["Qs"], g["Rs"], g["Fs"], g["goals"])]
        cost = terms[0]
        for t in terms[1:]:
            cost = cost + t
        return cost
    return QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"])


def indicator_cost_from_golden(g):
    """Controller cost of an indmppi_* fixture (gen_golden.gen_mppi_indicator) for the oracle: the sum,
    in the reference's order, of the quadratic / threshold / box terms the fixture holds."""
    terms = []
    if bool(g["has_quad"]):
        terms.append(QuadCostOracle(g["Q"], g["R"], g["F"], g["goal"]))
    if bool(g["has_thresh"]):
        terms.append(ThresholdCostOracle(g["goal"], g["thr_range"], float(g["thr"])))
    if bool(g["has_box"]):
        terms.append(BoxCostOracle(g["limits"]))
    return SumCostOracle(terms)


def hip_indicator_cost_from_golden(system, g):
    """The same cost as product objects."""
    from autompc_amd import BoxThresholdCost, QuadCost, ThresholdCost
    terms = []
    if bool(g["has_quad"]):
        terms.append(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    if bool(g["has_thresh"]):
        terms.append(ThresholdCost(system, g["goal"], [int(v) for v in g["thr_range"]], float(g["thr"])))
    if bool(g["has_box"]):
        terms.append(BoxThresholdCost(system, g["limits"], goal=g["goal"]))
    cost = terms[0]
    for t in terms[1:]:
        cost = cost + t
    return cost


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / scale
