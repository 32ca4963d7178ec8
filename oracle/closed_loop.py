"""Closed-loop surrogate simulation and trajectory score -- numpy restatement
(oracle, test-only).

reference: autompc/utils/simulation.py:11-64 (``simulate``) and the tuner's
score ``task.get_cost()(traj)`` (autompc/tuning/pipeline_tuner.py:223-231,
autompc/costs/cost.py:27-41).

simulate() appends the new observation with a ZERO control row after every
step, so a T-step run returns T+1 observations and T+1 control rows whose last
row is zero; the score sums stage costs over all T+1 rows and adds the terminal
cost of the last observation.
"""
import numpy as np


def simulate(controller, init_obs, sim_model, max_steps, traj_to_constate=None):
    nu = sim_model.system.ctrl_dim
    obs = [np.array(init_obs, dtype=np.float64)]
    ctrls = []
    if traj_to_constate is None:
        constate = np.concatenate([obs[0], np.zeros(nu)]) if controller.state_dim != sim_model.state_dim \
            else obs[0].copy()
    else:
        constate = traj_to_constate(obs[0])
    # sim_model.traj_to_state of the one-row trajectory (simulation.py:44-47): the observation
    # itself for the MLP, a lifted / stacked state for the linear models
    lift = getattr(sim_model, "state_from_first_obs", None)
    simstate = lift(obs[0]) if lift is not None else obs[0].copy()
    for _ in range(max_steps):
        u, constate = controller.run(constate, obs[-1])
        simstate = sim_model.pred(simstate, u)
        ctrls.append(np.array(u, dtype=np.float64))
        obs.append(simstate[:sim_model.system.obs_dim].copy())
    ctrls.append(np.zeros(nu))
    return np.array(obs), np.array(ctrls)
