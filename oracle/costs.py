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

    def traj_cost(self, obs, ctrls):
        """Cost.__call__ (cost.py:27-41): sum over ALL rows of obs and ctrl cost
        (the last row's ctrl is the zero row ``simulate`` appends) + terminal."""
        return (self.obs_cost_batch(obs).sum() + self.ctrl_cost_batch(ctrls).sum()
                + self.eval_term_obs_cost(obs[-1]))
