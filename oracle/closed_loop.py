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


def simulate(controller, init_obs, sim_model, max_steps, traj_to_constate=None, term_cond=None,
             dynamics=None):
    """term_cond(obs_rows, ctrl_rows) -> bool is asked after every step about the trajectory as
    simulate() holds it at that moment -- the new observation already appended, its control row
    still zero (simulation.py:59-63) -- and ends the run when true.  dynamics(obs, u) -> obs
    replaces the simulation model's step (simulation.py:54-58)."""
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
        if dynamics is None:
            simstate = sim_model.pred(simstate, u)
            nxt = simstate[:sim_model.system.obs_dim].copy()
        else:
            nxt = np.array(dynamics(obs[-1].copy(), u), dtype=np.float64)
        ctrls.append(np.array(u, dtype=np.float64))
        obs.append(nxt)
        if term_cond is not None and term_cond(np.array(obs), np.array(ctrls + [np.zeros(nu)])):
            break
    ctrls.append(np.zeros(nu))
    return np.array(obs), np.array(ctrls)


def num_steps_term_cond(num_steps):
    """Task.set_num_steps's termination condition (tasks/task.py:41-53): len(traj) >= num_steps."""
    return lambda obs_rows, ctrl_rows: len(obs_rows) >= num_steps


def eval_cfg_episode(controller, init_obs, sim_model, num_steps, traj_cost, term_cond=None,
                     dynamics=None, traj_to_constate=None):
    """One branch of PipelineTuner.eval_cfg (tuning/pipeline_tuner.py:222-233 surrogate,
    :244-251 true dynamics): reset the controller, simulate with the task's termination condition
    (the user's if one was set after set_num_steps, else len(traj) >= num_steps) capped at
    max_steps = num_steps, score with the task cost.  Returns (score, obs_rows, ctrl_rows)."""
    controller.reset()
    tc = term_cond if term_cond is not None else num_steps_term_cond(num_steps)
    obs, ctrls = simulate(controller, init_obs, sim_model, num_steps, traj_to_constate=traj_to_constate,
                          term_cond=tc, dynamics=dynamics)
    return traj_cost(obs, ctrls), obs, ctrls
