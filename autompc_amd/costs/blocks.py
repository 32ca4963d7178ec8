"""Turn a controller cost into the affine-quadratic block the device kernels evaluate
(include/autompc_hip.h: ampc_set_affine_quad_costs).

The reference's MPPI and iLQR call ``cost.eval_obs_cost`` / ``eval_ctrl_cost`` /
``eval_term_obs_cost`` (and their ``_hess`` variants) on whatever ``task.get_cost()`` returns
(autompc/control/mppi.py:73-82, ilqr.py:124-129,159-174).  For a ``SumCost`` those fan out term by
term (autompc/costs/sum_cost.py:49-54), so a sum of quadratic terms with DIFFERENT goals is a
perfectly good controller cost there -- it is what ``QuadCostFactory + GaussRegFactory`` builds
(gauss_reg_factory.py:37-45: ``Q = w inv(cov)``, goal = mean of the data).  About the first term's
goal ``g`` such a sum is

    stage     (x-g)'Q(x-g) + lin'(x-g) + c0 + u'Ru        Q = sum Q_k, R = sum R_k
    terminal  (x-g)'F(x-g) + lint'(x-g) + c1              F = sum F_k
    lin = sum (Q_k + Q_k')(g - g_k),  c0 = sum (g - g_k)'Q_k(g - g_k)     (lint, c1: with F_k)

which is one cost block of the kernels.  With a single term, or terms that share their goal, the
affine part is exactly zero and the block is the plain quadratic one.

Cost objects are recognised STRUCTURALLY (``.costs`` = a sum; quadratic leaf = ``is_quad`` with
``get_cost_matrices`` / ``get_goal``), never through a sum's own ``is_quad`` / ``get_goal``: the
reference's ``SumCost.get_goal`` returns a cost object instead of a vector (sum_cost.py:45-47), and
its ``is_quad`` is False as soon as two goals differ (sum_cost.py:84-93).  The reference's own cost
objects are therefore accepted as they are.
"""
import numpy as np


def _leaves(cost):
    subs = getattr(cost, "costs", None)
    if subs is not None and not callable(subs):
        out = []
        for c in subs:
            out.extend(_leaves(c))
        return out
    return [cost]


def is_quad_sum(cost):
    """True when `cost` is a quadratic cost or a (nested) sum of quadratic costs."""
    try:
        leaves = _leaves(cost)
    except Exception:
        return False
    if not leaves or not all(bool(getattr(c, "is_quad", False)) for c in leaves):
        return False
    # (terms that disagree on strict_reference have no single device block: quad_sum_block refuses them,
    #  so the controllers' is_compatible must not promise the pairing)
    return len({bool(getattr(c, "strict_reference", True)) for c in leaves}) == 1


def quad_sum_block(cost, obs_dim, ctrl_dim):
    """The device cost block of `cost`: dict with Q [no,no], R [nu,nu], F [no,no], goal [no],
    lin [no], lin_term [no], consts [2] and ``terminal_goal`` (False: iLQR's terminal gradient is
    the reference's goal-less one, cost.py:195,208-211; True: every term was built with
    ``strict_reference=False``).  TypeError for a term with no quadratic form."""
    no, nu = int(obs_dim), int(ctrl_dim)
    leaves = _leaves(cost)
    if not leaves:
        raise TypeError("empty cost")
    Q, R, F = np.zeros((no, no)), np.zeros((nu, nu)), np.zeros((no, no))
    lin, lint = np.zeros(no), np.zeros(no)
    c0 = c1 = 0.0
    g0 = None
    strict = []
    for c in leaves:
        if not getattr(c, "is_quad", False):
            raise TypeError("the HIP controllers evaluate sums of quadratic costs in-kernel; got a %s term"
                            % type(c).__name__)
        q, r, f = (np.asarray(m, dtype=np.float64) for m in c.get_cost_matrices())
        g = np.asarray(c.get_goal(), dtype=np.float64).reshape(no)
        if q.shape != (no, no) or f.shape != (no, no) or r.shape != (nu, nu):
            raise ValueError("cost matrices do not match the system (obs_dim %d, ctrl_dim %d)" % (no, nu))
        if g0 is None:
            g0 = g.copy()
        d = g0 - g
        Q += q
        R += r
        F += f
        lin += (q + q.T) @ d
        lint += (f + f.T) @ d
        c0 += d @ q @ d
        c1 += d @ f @ d
        strict.append(bool(getattr(c, "strict_reference", True)))
    if any(strict) and not all(strict):
        raise TypeError("terms of one controller cost must agree on strict_reference")
    return {"Q": Q, "R": R, "F": F, "goal": g0, "lin": lin, "lin_term": lint,
            "consts": np.array([c0, c1]), "terminal_goal": not strict[0]}


def _is_indicator(c):
    from .terms import _is_box, _is_threshold
    return _is_threshold(c) or _is_box(c)


def is_mppi_cost(cost):
    """True when `cost` is something the device MPPI evaluates in-kernel: a (nested) sum whose terms are
    quadratic costs and / or threshold / box indicators (thresh_cost.py:8-83; at most 8 of those) -- the
    reference's MPPI charges any Cost term by term (mppi.py:73-82)."""
    try:
        leaves = _leaves(cost)
    except Exception:
        return False
    quads = [c for c in leaves if getattr(c, "is_quad", False)]
    inds = [c for c in leaves if not getattr(c, "is_quad", False)]
    if not leaves or not all(_is_indicator(c) for c in inds) or len(inds) > 8:
        return False
    return len({bool(getattr(c, "strict_reference", True)) for c in quads}) <= 1


class _Terms:
    """The quadratic leaves of a cost as a sum (quad_sum_block walks `.costs`)."""
    def __init__(self, costs):
        self.costs = costs


def mppi_cost_parts(cost, obs_dim, ctrl_dim):
    """(block, terms): the affine-quadratic block of the quadratic terms of `cost` (all zeros when it
    has none) and the flattened threshold / box terms (kinds, params -- costs/terms.py's layout) or None.
    TypeError for a term of neither kind."""
    from .terms import cost_terms
    leaves = _leaves(cost)
    quads = [c for c in leaves if getattr(c, "is_quad", False)]
    inds = [c for c in leaves if not getattr(c, "is_quad", False)]
    for c in inds:
        if not _is_indicator(c):
            raise TypeError("the HIP MPPI evaluates quadratic, threshold and box cost terms in-kernel; got a %s term"
                            % type(c).__name__)
    no, nu = int(obs_dim), int(ctrl_dim)
    if quads:
        blk = quad_sum_block(_Terms(quads), no, nu)
    else:
        blk = {"Q": np.zeros((no, no)), "R": np.zeros((nu, nu)), "F": np.zeros((no, no)), "goal": np.zeros(no),
               "lin": np.zeros(no), "lin_term": np.zeros(no), "consts": np.zeros(2), "terminal_goal": False}
    terms = cost_terms(_Terms(inds), no, nu) if inds else None
    return blk, terms


def stack_blocks(blocks):
    """Blocks of several candidates -> the arrays Handle.set_cost_blocks takes."""
    return {k: np.stack([np.asarray(b[k], dtype=np.float64) for b in blocks])
            for k in ("Q", "R", "F", "goal", "lin", "lin_term", "consts")}
