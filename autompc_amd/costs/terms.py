"""Flatten a task cost into the term list the device scorer takes
(include/autompc_hip.h: ampc_score_trajectories).

A cost is a sum of terms (SumCost, reference autompc/costs/sum_cost.py:9-54) of three kinds:
quadratic (quad_cost.py:7-51), threshold and box indicators (thresh_cost.py:8-83).  Objects are
recognised structurally, so the reference's own cost objects flatten as well as this package's.
"""
import numpy as np

QUAD, THRESHOLD, BOX = 0, 1, 2


def _is_threshold(c):
    return hasattr(c, "_threshold") and hasattr(c, "_goal")


def _is_box(c):
    return hasattr(c, "_limits")


def _threshold_range(c):
    if hasattr(c, "_obs_range"):           # reference object
        return int(c._obs_range[0]), int(c._obs_range[1])
    return int(c._lo), int(c._hi)


def cost_terms(cost, obs_dim, ctrl_dim):
    """(kinds int32[n], params f64[...]) for `cost`; TypeError for a term with no device form."""
    kinds, params = [], []

    def visit(c):
        subs = getattr(c, "costs", None)
        if subs is not None and not callable(subs):
            for s in subs:
                visit(s)
            return
        if _is_threshold(c):
            lo, hi = _threshold_range(c)
            lo, hi = max(lo, 0), min(hi, obs_dim)      # numpy slicing clamps the same way
            if hi <= lo:
                # the reference's eval_obs_cost takes norm(., inf) of an EMPTY slice there: numpy raises
                raise ValueError("ThresholdCost with an empty obs_range %r: zero-size array to reduction operation "
                                 "maximum which has no identity" % (tuple(_threshold_range(c)),))
            goal = np.asarray(c._goal, dtype=np.float64).reshape(obs_dim)
            kinds.append(THRESHOLD)
            params.append(np.concatenate([goal, [lo, hi, float(c._threshold)]]))
        elif _is_box(c):
            lim = np.asarray(c._limits, dtype=np.float64).reshape(obs_dim, 2)
            kinds.append(BOX)
            params.append(np.concatenate([lim[:, 0], lim[:, 1]]))
        elif getattr(c, "is_quad", False):
            Q, R, F = (np.asarray(m, dtype=np.float64) for m in c.get_cost_matrices())
            goal = np.asarray(c.get_goal(), dtype=np.float64).reshape(obs_dim)
            kinds.append(QUAD)
            params.append(np.concatenate([Q.reshape(obs_dim * obs_dim), R.reshape(ctrl_dim * ctrl_dim),
                                          F.reshape(obs_dim * obs_dim), goal]))
        else:
            raise TypeError("no device form for cost term %s" % type(c).__name__)

    visit(cost)
    if not kinds:
        raise TypeError("empty cost")
    return np.asarray(kinds, dtype=np.int32), np.concatenate(params)
