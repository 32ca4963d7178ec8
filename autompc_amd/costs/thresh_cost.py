"""0/1 indicator costs used as *tuner scores* on the host
(reference: autompc/costs/thresh_cost.py:8-83).  They are never evaluated
inside the MPC solve (not differentiable, not quadratic); the closed-loop
evaluator applies them to the finished trajectory on the host.
"""
import numpy as np

from .cost import Cost


class ThresholdCost(Cost):
    def __init__(self, system, goal, obs_range, threshold):
        super().__init__(system)
        self._goal = np.array(goal, dtype=float)
        self._lo, self._hi = int(obs_range[0]), int(obs_range[1])
        self._threshold = float(threshold)
        self._has_goal = True

    def get_goal(self):
        return self._goal.copy()

    def eval_obs_cost(self, obs):
        dev = np.abs(np.asarray(obs)[self._lo:self._hi] - self._goal[self._lo:self._hi])
        return 1.0 if dev.size and dev.max() > self._threshold else 0.0

    def eval_ctrl_cost(self, ctrl):
        return 0.0

    def eval_term_obs_cost(self, obs):
        return 0.0


class BoxThresholdCost(Cost):
    def __init__(self, system, limits, goal=None):
        super().__init__(system)
        self._limits = np.array(limits, dtype=float)
        if goal is not None:
            self._goal = np.array(goal, dtype=float)
            self._has_goal = True

    def get_goal(self):
        if not self._has_goal:
            raise ValueError("Cost does not have goal")
        return self._goal.copy()

    def eval_obs_cost(self, obs):
        obs = np.asarray(obs)
        outside = np.any(obs < self._limits[:, 0]) or np.any(obs > self._limits[:, 1])
        return 1.0 if outside else 0.0

    def eval_ctrl_cost(self, ctrl):
        return 0.0

    def eval_term_obs_cost(self, obs):
        return 0.0
