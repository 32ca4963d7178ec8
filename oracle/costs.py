"""Quadratic cost terms -- numpy restatement (oracle, test-only).

reference: autompc/costs/cost.py:66-83 (obs), :118-134 (ctrl), :166-183 (term),
:85-116 / :136-164 (grad/hess), :185-213 (terminal grad/hess -- these use ``obs``
and NOT ``obs - goal``: kept).
"""
import numpy as np


class QuadCostOracle:
    def __init__(self, Q, R, F, goal):
        self.Q = np.asarray(Q, dtype=np.float64)
        self.R = np.asarray(R, dtype=np.float64)
        self.F = np.asarray(F, dtype=np.float64)
        self.goal = np.asarray(goal, dtype=np.float64)

    @classmethod
    def from_cost(cls, cost):
        Q, R, F = cost.get_cost_matrices()
        return cls(Q, R, F, cost.get_goal())

    # scalar entry points (what the reference calls 2*N*H times per MPPI solve)
    def eval_obs_cost(self, obs):
        d = obs - self.goal
        return d.T @ self.Q @ d

    def eval_ctrl_cost(self, ctrl):
        return ctrl.T @ self.R @ ctrl

    def eval_term_obs_cost(self, obs):
        d = obs - self.goal
        return d.T @ self.F @ d

    def eval_obs_cost_hess(self, obs):
        d = obs - self.goal
        S = self.Q + self.Q.T
        return d.T @ self.Q @ d, S @ d, S

    def eval_ctrl_cost_hess(self, ctrl):
        S = self.R + self.R.T
        return ctrl.T @ self.R @ ctrl, S @ ctrl, S

    def eval_term_obs_cost_hess(self, obs):
        S = self.F + self.F.T
        return obs.T @ self.F @ obs, S @ obs, S

    # vectorised forms (same arithmetic, batched) used by the fast oracle mode
    def obs_cost_batch(self, obs):
        d = obs - self.goal
        return np.einsum("ni,ij,nj->n", d, self.Q, d)

    def ctrl_cost_batch(self, ctrls):
        return np.einsum("ni,ij,nj->n", ctrls, self.R, ctrls)

    def term_cost_batch(self, obs):
        d = obs - self.goal
        return np.einsum("ni,ij,nj->n", d, self.F, d)

    def traj_cost(self, obs, ctrls):
        """Cost.__call__ (cost.py:27-41): sum over ALL rows of obs and ctrl cost
        (the last row's ctrl is the zero row ``simulate`` appends) + terminal."""
        return (self.obs_cost_batch(obs).sum() + self.ctrl_cost_batch(ctrls).sum()
                + self.eval_term_obs_cost(obs[-1]))


class SumCostOracle:
    """Sum of quadratic terms -- SumCost._sum_results (sum_cost.py:49-54): every ``eval_*`` call is
    fanned out to the terms IN ORDER and the results (scalars, or tuples entry by entry) are added
    with Python's ``sum``.  The goals of the terms may differ (QuadCostFactory + GaussRegFactory,
    gauss_reg_factory.py:37-45); the terminal ``_hess`` of every term ignores that term's goal
    (cost.py:195,208-211), so the sum's does too."""

    def __init__(self, terms):
        self.terms = list(terms)

    @classmethod
    def from_arrays(cls, Qs, Rs, Fs, goals):
        return cls([QuadCostOracle(q, r, f, g) for q, r, f, g in zip(Qs, Rs, Fs, goals)])

    def _fan(self, attr, arg):
        res = [getattr(t, attr)(arg) for t in self.terms]
        if isinstance(res[0], tuple):
            return tuple(sum(v) for v in zip(*res))
        return sum(res)

    def eval_obs_cost(self, obs):
        return self._fan("eval_obs_cost", obs)

    def eval_ctrl_cost(self, ctrl):
        return self._fan("eval_ctrl_cost", ctrl)

    def eval_term_obs_cost(self, obs):
        return self._fan("eval_term_obs_cost", obs)

    def eval_obs_cost_hess(self, obs):
        return self._fan("eval_obs_cost_hess", obs)

    def eval_ctrl_cost_hess(self, ctrl):
        return self._fan("eval_ctrl_cost_hess", ctrl)

    def eval_term_obs_cost_hess(self, obs):
        return self._fan("eval_term_obs_cost_hess", obs)

    def obs_cost_batch(self, obs):
        return sum(t.obs_cost_batch(obs) for t in self.terms)

    def ctrl_cost_batch(self, ctrls):
        return sum(t.ctrl_cost_batch(ctrls) for t in self.terms)

    def term_cost_batch(self, obs):
        return sum(t.term_cost_batch(obs) for t in self.terms)

    def traj_cost(self, obs, ctrls):
        return sum(t.traj_cost(obs, ctrls) for t in self.terms)


class _IndicatorOracle:
    """A cost term that is 1 for every row whose observation violates a condition and has no control or
    terminal part (thresh_cost.py:34-38, 79-83)."""

    def violated(self, obs):
        raise NotImplementedError

    def eval_obs_cost(self, obs):
        return 1.0 if self.violated(np.asarray(obs, dtype=np.float64)) else 0.0

    def eval_ctrl_cost(self, ctrl):
        return 0.0

    def eval_term_obs_cost(self, obs):
        return 0.0

    def obs_cost_batch(self, obs):
        return np.array([self.eval_obs_cost(o) for o in np.asarray(obs, dtype=np.float64)])

    def ctrl_cost_batch(self, ctrls):
        return np.zeros(np.asarray(ctrls).shape[0])

    def term_cost_batch(self, obs):
        return np.zeros(np.asarray(obs).shape[0])

    def traj_cost(self, obs, ctrls):
        return float(self.obs_cost_batch(obs).sum())


class ThresholdCostOracle(_IndicatorOracle):
    """ThresholdCost.eval_obs_cost (thresh_cost.py:27-32): 1 when the infinity norm of (obs - goal) over
    obs_range[0] <= i < obs_range[1] exceeds the threshold."""

    def __init__(self, goal, obs_range, threshold):
        self.goal, self.thr = np.asarray(goal, dtype=np.float64), float(threshold)
        self.lo, self.hi = int(obs_range[0]), int(obs_range[1])

    def violated(self, obs):
        gap = np.abs(obs[self.lo:self.hi] - self.goal[self.lo:self.hi])
        return bool(np.max(gap) > self.thr)      # (numpy: NaN if any entry is NaN -> False; ValueError on an empty range)


class BoxCostOracle(_IndicatorOracle):
    """BoxThresholdCost.eval_obs_cost (thresh_cost.py:73-77): 1 when some obs_i leaves [lower_i, upper_i]."""

    def __init__(self, limits):
        self.limits = np.asarray(limits, dtype=np.float64)

    def violated(self, obs):
        return bool((obs < self.limits[:, 0]).any() or (obs > self.limits[:, 1]).any())


def score_terms(kinds, params, obs, ctrls, obs_dim=None):
    """Cost.__call__ (cost.py:27-41) of ONE trajectory obs [T,ns], ctrls [T,nu] under a flattened
    sum of terms (layout of include/autompc_hip.h: ampc_score_trajectories): row by row, term by
    term, exactly as SumCost._sum_results (sum_cost.py:49-54) fans the calls out.
      0 quad       (quad_cost.py:7-51)   1 threshold (thresh_cost.py:27-32)
      2 box        (thresh_cost.py:73-77)"""
    obs, ctrls = np.asarray(obs, dtype=np.float64), np.asarray(ctrls, dtype=np.float64)
    no = obs.shape[1] if obs_dim is None else obs_dim
    nu = ctrls.shape[1]
    terms, o = [], 0
    for k in kinds:
        if k == 0:
            n = 2 * no * no + nu * nu + no
            p = params[o:o + n]
            terms.append((0, QuadCostOracle(p[:no * no].reshape(no, no),
                                            p[no * no:no * no + nu * nu].reshape(nu, nu),
                                            p[no * no + nu * nu:2 * no * no + nu * nu].reshape(no, no),
                                            p[2 * no * no + nu * nu:])))
        elif k == 1:
            n = no + 3
            p = params[o:o + n]
            terms.append((1, (p[:no], int(p[no]), int(p[no + 1]), p[no + 2])))
        elif k == 2:
            n = 2 * no
            p = params[o:o + n]
            terms.append((2, (p[:no], p[no:])))
        else:
            raise ValueError("unknown term kind %r" % (k,))
        o += n
    total = 0.0
    for t in range(obs.shape[0]):
        x, u = obs[t, :no], ctrls[t]
        for kind, d in terms:
            if kind == 0:
                total += d.eval_obs_cost(x) + d.eval_ctrl_cost(u)
            elif kind == 1:
                goal, lo, hi, thr = d
                dev = np.abs(x[lo:hi] - goal[lo:hi])
                total += 1.0 if dev.size and dev.max() > thr else 0.0
            else:
                lo, hi = d
                total += 1.0 if ((x < lo).any() or (x > hi).any()) else 0.0
    for kind, d in terms:
        if kind == 0:
            total += d.eval_term_obs_cost(obs[-1, :no])
    return total
