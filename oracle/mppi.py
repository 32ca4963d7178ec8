"""MPPI solve -- numpy restatement (oracle, test-only).

Follows autompc/control/mppi.py:
  :16-24   noise = np.random.normal(scale=sqrt(sigma), size=shape+(1,)) from the
           GLOBAL legacy numpy stream                      -> _draw()
  :97-99   act_sequence is a random draw of shape (H,1) at construction / reset
  :110-118 update(): softmin weights, weighted noise sum   -> update()
  :120-152 do_rollouts(): shift, sample, clip, accumulate  -> do_rollouts()
  :154-168 run()                                           -> run()

Generalisation (the reference only works for ctrl_dim == 1): the
noise has trailing dimension ``nu`` instead of the hard-coded 1.  With nu == 1
every array shape, draw order and arithmetic step is the reference's.

Kept quirks: the "terminal cost" is the scalar terminal cost of the LAST
particle added to every particle (mppi.py:79-82,146-148); stage
costs carry no dt; ``lmda / sigma`` scales the action cost where sigma is the
noise VARIANCE; controls are optimised in units of ``umax`` (ctrl_scale).

``strict_reference=True`` evaluates the stage cost with the reference's
per-particle Python loop (mppi.py:73-78) -- that is the structure the CPU
baseline is timed on; ``False`` uses the batched einsum (same values to
rounding, used to keep the parity tests fast).
"""
import numpy as np


class MPPIOracle:
    def __init__(self, model, cost, ctrl_bounds, horizon=20, num_path=1000, sigma=1.0,
                 lmda=1.0, strict_reference=False, per_particle_terminal=False):
        self.model = model
        self.cost = cost
        self.nx = model.state_dim
        self.obs_dim = model.system.obs_dim
        self.nu = model.system.ctrl_dim
        self.H = int(horizon)
        self.num_path = int(num_path)
        self.sigma = sigma
        self.lmda = lmda
        self.strict_reference = strict_reference
        self.per_particle_terminal = per_particle_terminal
        bounds = np.asarray(ctrl_bounds, dtype=np.float64)
        self.umin = bounds[:, 0].copy()
        self.umax = bounds[:, 1].copy()
        self.ctrl_scale = self.umax
        self.scale = np.sqrt(sigma)
        self.reset()

    def _draw(self, shape):
        return np.random.normal(scale=self.scale, size=tuple(shape) + (self.nu,))

    def reset(self):
        self.act_sequence = self._draw((self.H,))
        self.cur_step = 0

    # -- stage cost ---------------------------------------------------------
    def _stage_cost(self, path, ctrls):
        if self.strict_reference:
            out = np.zeros(path.shape[0])
            for i in range(path.shape[0]):
                out[i] += self.cost.eval_obs_cost(path[i, :self.obs_dim])
                out[i] += self.cost.eval_ctrl_cost(ctrls[i, :])
            return out
        return (self.cost.obs_cost_batch(path[:, :self.obs_dim])
                + self.cost.ctrl_cost_batch(ctrls))

    # -- one sampling pass --------------------------------------------------
    def do_rollouts(self, x0, eps_nhu=None):
        a = self.act_sequence
        a[:-1] = a[1:]
        a[-1] = a[-2]
        if eps_nhu is None:
            eps_nhu = self._draw((self.num_path, self.H))
        eps = np.array(eps_nhu, dtype=np.float64).transpose((1, 0, 2))  # (H, N, nu) view
        path = np.tile(np.asarray(x0, dtype=np.float64), (self.num_path, 1))
        costs = np.zeros(self.num_path)
        action_cost = np.zeros(self.num_path)
        lo = self.umin / self.ctrl_scale
        hi = self.umax / self.ctrl_scale
        for i in range(self.H):
            actions = np.minimum(hi, np.maximum(lo, eps[i] + a[i]))
            eps[i] = actions - a[i]
            scaled = actions * self.ctrl_scale
            costs += self._stage_cost(path, scaled)
            action_cost += self.lmda / self.sigma * np.einsum("ij,ij->i", actions, eps[i])
            path = self.model.pred_batch(path, scaled)
        if self.per_particle_terminal:
            costs += self.cost.term_cost_batch(path[:, :self.obs_dim])
        else:
            costs += self.cost.eval_term_obs_cost(path[-1, :self.obs_dim])
        costs += action_cost
        self.last_path = path
        return costs, eps

    def update(self, costs, eps):
        S = np.exp(-1.0 / self.lmda * (costs - np.amin(costs)))
        weight = S / np.sum(S)
        self.act_sequence += np.sum(eps * weight[None, :, None], axis=1)
        return weight

    def run(self, constate, new_obs, eps_nhu=None):
        x0 = self.model.update_state(constate[:-self.nu], constate[-self.nu:], new_obs)
        costs, eps = self.do_rollouts(x0, eps_nhu)
        self.update(costs, eps)
        self.cur_step += 1
        self.last_costs, self.last_eps = costs, eps
        u = self.act_sequence[0].copy() * self.ctrl_scale
        return u, np.concatenate([x0, u])

    def traj_to_state(self, traj):
        return np.concatenate([self.model.traj_to_state(traj), traj[-1].ctrl])

    @property
    def state_dim(self):
        return self.nx + self.nu
