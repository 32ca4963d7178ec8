"""Stage / terminal cost interface and the quadratic cost the HIP kernels evaluate.

The method names and return conventions follow the reference's ``Cost`` ABC
(reference: autompc/costs/cost.py:27-41 ``__call__``, :43-64 matrices / goal,
:66-213 the nine ``eval_*`` entry points) because MPPI/iLQR callers and the
tuner score (``task.get_cost()(traj)``) are written against them.

Quirk kept on purpose ("bug-compatibility"): the reference's
``eval_term_obs_cost_diff`` / ``_hess`` differentiate ``obs' F obs`` and ignore
the goal (cost.py:195, :208-211).  iLQR's terminal value function is seeded
from those, so the default here (``strict_reference=True``) reproduces it;
``strict_reference=False`` uses ``obs - goal`` consistently.
"""
import numpy as np


class Cost:
    def __init__(self, system):
        self.system = system
        self._is_quad = False
        self._is_convex = False
        self._is_diff = False
        self._is_twice_diff = False
        self._has_goal = False

    # whole-trajectory score, as the tuner uses it (cost.py:27-41)
    def __call__(self, traj):
        total = 0.0
        for i in range(len(traj)):
            step = traj[i]
            total += self.eval_obs_cost(step.obs)
            total += self.eval_ctrl_cost(step.ctrl)
        total += self.eval_term_obs_cost(traj[-1].obs)
        return total

    def get_cost_matrices(self):
        raise ValueError("Cost is not quadratic.")

    def get_goal(self):
        raise ValueError("Cost does not have goal")

    def eval_obs_cost(self, obs):
        raise NotImplementedError

    def eval_obs_cost_diff(self, obs):
        raise NotImplementedError

    def eval_obs_cost_hess(self, obs):
        raise NotImplementedError

    def eval_ctrl_cost(self, ctrl):
        raise NotImplementedError

    def eval_ctrl_cost_diff(self, ctrl):
        raise NotImplementedError

    def eval_ctrl_cost_hess(self, ctrl):
        raise NotImplementedError

    def eval_term_obs_cost(self, obs):
        raise NotImplementedError

    def eval_term_obs_cost_diff(self, obs):
        raise NotImplementedError

    def eval_term_obs_cost_hess(self, obs):
        raise NotImplementedError

    @property
    def is_quad(self):
        return self._is_quad

    @property
    def is_convex(self):
        return self._is_convex

    @property
    def is_diff(self):
        return self._is_diff

    @property
    def is_twice_diff(self):
        return self._is_twice_diff

    @property
    def has_goal(self):
        return self._has_goal

    def __add__(self, other):
        from .sum_cost import SumCost
        if isinstance(other, SumCost):
            return other.__radd__(self)
        return SumCost(self.system, [self, other])


class QuadCost(Cost):
    """(x-g)' Q (x-g) + u' R u per stage, (x-g)' F (x-g) at the end
    (reference: autompc/costs/quad_cost.py:7-51)."""

    def __init__(self, system, Q, R, F=None, goal=None, strict_reference=True):
        super().__init__(system)
        no, nu = system.obs_dim, system.ctrl_dim
        Q = np.asarray(Q, dtype=float)
        R = np.asarray(R, dtype=float)
        if Q.shape != (no, no):
            raise ValueError("Q is the wrong shape")
        if R.shape != (nu, nu):
            raise ValueError("R is the wrong shape")
        if F is None:
            F = np.zeros((no, no))
        F = np.asarray(F, dtype=float)
        if F.shape != (no, no):
            raise ValueError("F is the wrong shape")
        self._Q, self._R, self._F = Q.copy(), R.copy(), F.copy()
        self._goal = np.zeros(no) if goal is None else np.array(goal, dtype=float)
        self.strict_reference = bool(strict_reference)
        self._is_quad = self._is_convex = self._is_diff = True
        self._is_twice_diff = self._has_goal = True

    def get_cost_matrices(self):
        return self._Q.copy(), self._R.copy(), self._F.copy()

    def get_goal(self):
        return self._goal.copy()

    @staticmethod
    def _form(M, v):
        return v @ M @ v

    def eval_obs_cost(self, obs):
        return self._form(self._Q, obs - self._goal)

    def eval_obs_cost_diff(self, obs):
        d = obs - self._goal
        return self._form(self._Q, d), (self._Q + self._Q.T) @ d

    def eval_obs_cost_hess(self, obs):
        d = obs - self._goal
        sym = self._Q + self._Q.T
        return self._form(self._Q, d), sym @ d, sym

    def eval_ctrl_cost(self, ctrl):
        return self._form(self._R, ctrl)

    def eval_ctrl_cost_diff(self, ctrl):
        return self._form(self._R, ctrl), (self._R + self._R.T) @ ctrl

    def eval_ctrl_cost_hess(self, ctrl):
        sym = self._R + self._R.T
        return self._form(self._R, ctrl), sym @ ctrl, sym

    def eval_term_obs_cost(self, obs):
        return self._form(self._F, obs - self._goal)

    def _term_point(self, obs):
        return obs if self.strict_reference else obs - self._goal

    def eval_term_obs_cost_diff(self, obs):
        d = self._term_point(obs)
        return self._form(self._F, d), (self._F + self._F.T) @ d

    def eval_term_obs_cost_hess(self, obs):
        d = self._term_point(obs)
        sym = self._F + self._F.T
        return self._form(self._F, d), sym @ d, sym
