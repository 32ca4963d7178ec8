"""Sum of cost terms (reference: autompc/costs/sum_cost.py:9-138).

``a + b`` on two ``Cost`` objects builds one of these.  A sum of quadratic
terms that share a goal is itself quadratic (``is_quad``), and
``get_cost_matrices`` returns the summed Q, R, F -- which is what the HIP
kernels are handed, so a SumCost-of-QuadCost is evaluated in-kernel as a single
quadratic form.
"""
from collections.abc import Iterable

import numpy as np

from .cost import Cost


class SumCost(Cost):
    def __init__(self, system, costs):
        super().__init__(system)
        self._costs = list(costs)

    @property
    def costs(self):
        return list(self._costs)

    def _all(self, flag):
        return all(getattr(c, flag) for c in self._costs)

    def _shared_goal(self):
        first = self._costs[0]
        if not first.has_goal:
            return False
        g = first.get_goal()
        return all(c.has_goal and np.array_equal(g, c.get_goal()) for c in self._costs[1:])

    @property
    def is_quad(self):
        return self._all("is_quad") and self._shared_goal()

    @property
    def is_convex(self):
        return self._all("is_convex")

    @property
    def is_diff(self):
        return self._all("is_diff")

    @property
    def is_twice_diff(self):
        return self._all("is_diff")

    @property
    def has_goal(self):
        return self._shared_goal()

    def get_cost_matrices(self):
        if not self.is_quad:
            raise NotImplementedError
        no, nu = self.system.obs_dim, self.system.ctrl_dim
        Q, R, F = np.zeros((no, no)), np.zeros((nu, nu)), np.zeros((no, no))
        for c in self._costs:
            q, r, f = c.get_cost_matrices()
            Q += q
            R += r
            F += f
        return Q, R, F

    def get_goal(self):
        # NOTE: the reference returns ``self.costs[0]`` (the first *cost object*,
        # sum_cost.py:45-47), which is unusable as a goal vector; the shared goal
        # is returned here instead.
        if self.has_goal:
            return self._costs[0].get_goal()
        raise ValueError("Cost does not have goal")

    def _fan_out(self, method, arg):
        parts = [getattr(c, method)(arg) for c in self._costs]
        if isinstance(parts[0], Iterable):
            return [sum(col) for col in zip(*parts)]
        return sum(parts)

    def eval_obs_cost(self, obs):
        return self._fan_out("eval_obs_cost", obs)

    def eval_obs_cost_diff(self, obs):
        return self._fan_out("eval_obs_cost_diff", obs)

    def eval_obs_cost_hess(self, obs):
        return self._fan_out("eval_obs_cost_hess", obs)

    def eval_ctrl_cost(self, ctrl):
        return self._fan_out("eval_ctrl_cost", ctrl)

    def eval_ctrl_cost_diff(self, ctrl):
        return self._fan_out("eval_ctrl_cost_diff", ctrl)

    def eval_ctrl_cost_hess(self, ctrl):
        return self._fan_out("eval_ctrl_cost_hess", ctrl)

    def eval_term_obs_cost(self, obs):
        return self._fan_out("eval_term_obs_cost", obs)

    def eval_term_obs_cost_diff(self, obs):
        return self._fan_out("eval_term_obs_cost_diff", obs)

    def eval_term_obs_cost_hess(self, obs):
        return self._fan_out("eval_term_obs_cost_hess", obs)

    def __add__(self, other):
        extra = other.costs if isinstance(other, SumCost) else [other]
        return SumCost(self.system, self._costs + list(extra))

    def __radd__(self, other):
        first = other.costs if isinstance(other, SumCost) else [other]
        return SumCost(self.system, list(first) + self._costs)
