"""Closed-loop driver: ``simulate(controller, init_obs, ...)`` with the reference's call signature
and return value (reference: autompc/utils/simulation.py:11-64).

The loop is the plain host-side version -- one ``controller.run`` per step, for a single
controller.  The device-resident, batched counterpart is
``autompc_amd.tuning.CandidateEvaluator`` / ``ampc_mppi_closed_loop``.
"""
import numpy as np

from .trajectory import Trajectory


def simulate(controller, init_obs, term_cond=None, dynamics=None, sim_model=None,
             max_steps=10000, silent=True):
    """Roll `controller` out from `init_obs` for at most `max_steps` steps against either the
    true `dynamics(obs, ctrl) -> obs` or a `sim_model` (its own state is carried along and the
    observation is the first obs_dim entries of it).  Returns the Trajectory of T+1 rows the
    reference returns: row t holds obs_t and the control applied there, the final row's control
    is zero."""
    if sim_model is None and dynamics is None:
        raise ValueError("Must specify dynamics function or simulation model")
    system = controller.system
    no, nu = system.obs_dim, system.ctrl_dim
    obs_rows = np.zeros((max_steps + 1, no)) if max_steps <= 100000 else None
    ctrl_rows = np.zeros((max_steps + 1, nu)) if max_steps <= 100000 else None

    def grown(buf, rows, width):
        # amortised growth for open-ended runs (max_steps is only a cap)
        if buf is not None and rows <= buf.shape[0]:
            return buf
        bigger = np.zeros((max(2 * rows, 1024), width))
        if buf is not None:
            bigger[:buf.shape[0]] = buf
        return bigger

    obs_rows = grown(obs_rows, 1, no)
    ctrl_rows = grown(ctrl_rows, 1, nu)
    obs_rows[0] = np.asarray(init_obs, dtype=np.float64)
    so_far = Trajectory(system, 1, obs_rows[:1], ctrl_rows[:1])
    ctl_state = controller.traj_to_state(so_far)
    model_state = sim_model.traj_to_state(so_far) if dynamics is None else None
    n = 0                                   # completed steps; rows 0..n are valid
    while n < max_steps:
        u, ctl_state = controller.run(ctl_state, obs_rows[n].copy())
        ctrl_rows[n] = u
        if dynamics is not None:
            nxt = dynamics(obs_rows[n].copy(), u)
        else:
            model_state = sim_model.pred(model_state, u)
            nxt = model_state[:no]
        n += 1
        obs_rows = grown(obs_rows, n + 1, no)
        ctrl_rows = grown(ctrl_rows, n + 1, nu)
        obs_rows[n] = nxt
        ctrl_rows[n] = 0.0
        if term_cond is not None and term_cond(Trajectory(system, n + 1, obs_rows[:n + 1],
                                                          ctrl_rows[:n + 1])):
            break
    return Trajectory(system, n + 1, obs_rows[:n + 1].copy(), ctrl_rows[:n + 1].copy())
