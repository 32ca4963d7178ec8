"""0/1 indicator costs: the scores the benchmark tasks are tuned on
(reference: autompc/costs/thresh_cost.py:8-83).

They count the time steps at which the observation violates a condition; there is no control
or terminal part and no derivative, so they never enter an MPC solve.  As *task scores* they
are evaluated on the device for whole batches of trajectories (``autompc_amd.costs.cost_terms``
-> ``ampc_score_trajectories``); the methods below are the per-row host interface every ``Cost``
has.
"""
import numpy as np

from .cost import Cost


class _IndicatorCost(Cost):
    """1.0 for a row whose observation `_violated`, else 0.0."""

    def _violated(self, obs):
        raise NotImplementedError

    def eval_obs_cost(self, obs):
        return float(bool(self._violated(np.asarray(obs, dtype=float))))

    def eval_ctrl_cost(self, ctrl):
        return 0.0

    def eval_term_obs_cost(self, obs):
        return 0.0

    def get_goal(self):
        if not self._has_goal:
            raise ValueError("Cost does not have goal")
        return self._goal.copy()


class ThresholdCost(_IndicatorCost):
    """Violated when max_i |obs_i - goal_i| > threshold over obs_range[0] <= i < obs_range[1]."""

    def __init__(self, system, goal, obs_range, threshold):
        super().__init__(system)
        self._goal = np.array(goal, dtype=float)
        self._has_goal = True
        self._lo, self._hi = (int(v) for v in obs_range[:2])
        self._threshold = float(threshold)

    def _violated(self, obs):
        window = slice(self._lo, self._hi)
        gap = np.abs(obs[window] - self._goal[window])
        return gap.size > 0 and gap.max() > self._threshold


class BoxThresholdCost(_IndicatorCost):
    """Violated when any obs_i leaves [limits[i, 0], limits[i, 1]] (use +-inf for open sides).
    `goal` is not used by the cost itself; cost factories downstream may read it."""

    def __init__(self, system, limits, goal=None):
        super().__init__(system)
        self._limits = np.array(limits, dtype=float)
        if goal is not None:
            self._goal = np.array(goal, dtype=float)
            self._has_goal = True

    def _violated(self, obs):
        below, above = obs < self._limits[:, 0], obs > self._limits[:, 1]
        return bool(below.any() or above.any())
