"""Control task: cost + control bounds + initial observation + episode length.

Only the part of the reference's ``autompc.Task`` that the MPC hot path and
the closed-loop evaluator read is reproduced (reference:
autompc/tasks/task.py:42-71 num_steps, :103-146 cost / init_obs,
:182-267 control bounds).  State constraints and equality/inequality
constraint lists are consumed only by the reference's NMPC controller, which
is out of scope.
"""
import numpy as np


class Task:
    def __init__(self, system):
        self.system = system
        self._ctrl_bounds = np.empty((system.ctrl_dim, 2))
        self._ctrl_bounds[:, 0] = -np.inf
        self._ctrl_bounds[:, 1] = np.inf
        self._obs_bounds = np.empty((system.obs_dim, 2))
        self._obs_bounds[:, 0] = -np.inf
        self._obs_bounds[:, 1] = np.inf
        self._init_obs = None
        self._num_steps = None
        self._term_cond = None
        self.cost = None

    # -- episode length / termination ---------------------------------------
    def set_num_steps(self, num_steps):
        self._num_steps = int(num_steps)
        self._term_cond = None

    def has_num_steps(self):
        return self._num_steps is not None

    def get_num_steps(self):
        return self._num_steps

    def set_term_cond(self, term_cond):
        self._term_cond = term_cond

    def has_user_term_cond(self):
        """True when set_term_cond installed a condition after the last set_num_steps (the
        reference keeps both in one slot, tasks/task.py:52,101; the batched evaluator needs to know
        whether the episode length is known up front)."""
        return self._term_cond is not None

    def term_cond(self, traj):
        if self._term_cond is not None:
            return bool(self._term_cond(traj))
        if self._num_steps is not None:
            return len(traj) >= self._num_steps
        return False

    # -- cost / initial observation -----------------------------------------
    def set_cost(self, cost):
        self.cost = cost

    def get_cost(self):
        return self.cost

    def set_init_obs(self, init_obs):
        self._init_obs = np.array(init_obs, dtype=float)

    def get_init_obs(self):
        return None if self._init_obs is None else self._init_obs.copy()

    # -- bounds ---------------------------------------------------------------
    def set_ctrl_bound(self, ctrl_label, lower, upper):
        self._ctrl_bounds[self.system.controls.index(ctrl_label), :] = (lower, upper)

    def set_ctrl_bounds(self, lowers, uppers):
        self._ctrl_bounds[:, 0] = lowers
        self._ctrl_bounds[:, 1] = uppers

    def get_ctrl_bounds(self):
        return self._ctrl_bounds.copy()

    def are_ctrl_bounded(self):
        return bool(np.any(np.isfinite(self._ctrl_bounds)))

    def set_obs_bound(self, obs_label, lower, upper):
        self._obs_bounds[self.system.observations.index(obs_label), :] = (lower, upper)

    def set_obs_bounds(self, lowers, uppers):
        self._obs_bounds[:, 0] = lowers
        self._obs_bounds[:, 1] = uppers

    def get_obs_bounds(self):
        return self._obs_bounds.copy()

    def are_obs_bounded(self):
        return bool(np.any(np.isfinite(self._obs_bounds)))

    def is_cost_quad(self):
        return bool(getattr(self.cost, "is_quad", False))
