"""Closed-form nonlinear test model with exact Jacobians (oracle, test-only).

Plays the role the reference's ``DummyNonlinear`` double was meant to play
(autompc/sysid/dummy_nonlinear.py:13-51 is not usable as shipped, SURVEY.md
section 4): a tiny differentiable model for exercising iLQR without an MLP.

    x0' = x0 + a*x1 + b*x1^3
    x1' = x1 + c*u - d*sin(x0)
"""
import numpy as np


class CubicIntegrator:
    a, b, c, d = 0.05, 0.1, 0.1, 0.02

    def __init__(self, system):
        if system.obs_dim != 2 or system.ctrl_dim != 1:
            raise ValueError("CubicIntegrator is a 2-state / 1-control model")
        self.system = system

    @property
    def state_dim(self):
        return 2

    def traj_to_state(self, traj):
        return traj[-1].obs.copy()

    def update_state(self, state, new_ctrl, new_obs):
        return np.array(new_obs, dtype=np.float64)

    def pred_batch(self, states, ctrls):
        x0, x1, u = states[:, 0], states[:, 1], ctrls[:, 0]
        out = np.empty_like(states)
        out[:, 0] = x0 + self.a * x1 + self.b * x1 ** 3
        out[:, 1] = x1 + self.c * u - self.d * np.sin(x0)
        return out

    def pred(self, state, ctrl):
        return self.pred_batch(state[None, :], ctrl[None, :])[0]

    def pred_diff_batch(self, states, ctrls):
        m = states.shape[0]
        jx = np.zeros((m, 2, 2))
        ju = np.zeros((m, 2, 1))
        jx[:, 0, 0] = 1.0
        jx[:, 0, 1] = self.a + 3.0 * self.b * states[:, 1] ** 2
        jx[:, 1, 0] = -self.d * np.cos(states[:, 0])
        jx[:, 1, 1] = 1.0
        ju[:, 1, 0] = self.c
        return self.pred_batch(states, ctrls), jx, ju

    def pred_diff(self, state, ctrl):
        o, a, b = self.pred_diff_batch(state[None, :], ctrl[None, :])
        return o[0], a[0], b[0]
