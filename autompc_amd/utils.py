"""Closed-loop driver with the reference's ``simulate`` signature
(reference: autompc/utils/simulation.py:11-64)."""
import numpy as np

from .trajectory import extend, zeros


def simulate(controller, init_obs, term_cond=None, dynamics=None, sim_model=None,
             max_steps=10000, silent=True):
    if dynamics is None and sim_model is None:
        raise ValueError("Must specify dynamics function or simulation model")
    system = controller.system
    traj = zeros(system, 1)
    x = np.array(init_obs, dtype=np.float64)
    traj.obs[0, :] = x
    constate = controller.traj_to_state(traj)
    simstate = sim_model.traj_to_state(traj) if dynamics is None else None
    for _ in range(max_steps):
        u, constate = controller.run(constate, traj[-1].obs)
        if dynamics is None:
            simstate = sim_model.pred(simstate, u)
            x = simstate[:system.obs_dim]
        else:
            x = dynamics(x, u)
        traj.ctrls[-1, :] = u
        traj = extend(traj, [x], np.zeros((1, system.ctrl_dim)))
        if term_cond is not None and term_cond(traj):
            break
    return traj
