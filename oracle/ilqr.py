"""iLQR solve -- numpy restatement (oracle, test-only).

Follows autompc/control/ilqr.py:
  :124-129  objective: dt * sum(stage costs) + terminal cost      -> _objective()
  :141-149  initial rollout with per-step pred_diff                -> solve() prologue
  :159-187  backward Riccati sweep, unregularised solve            -> _backward()
  :196-205  all ls_max_iter step sizes rolled out together         -> _line_search_rollout()
  :207-236  acceptance / best-so-far bookkeeping / Jacobian refresh
  :236-261  failure test, ||du|| convergence test, swap
  :267-295  run(): full re-solve from a zero guess every control step

Constants: u_threshold 1e-3, max_iter 50, ls_max_iter 10, ls_discount 0.2,
ls_cost_threshold 0.3 (ilqr.py:100-101).

Kept quirks: terminal gradient/Hessian ignore the goal (cost.py:195,208-211);
when the line search neither succeeds nor trips the failure test, the LAST
candidate evaluated becomes the new nominal trajectory while the Jacobians are
left stale (Python loop-variable leak at ilqr.py:208-255); ``np.linalg.solve``
is unregularised, a singular Quu raises LinAlgError.
"""
import numpy as np


class ILQROracle:
    def __init__(self, model, cost, dt, horizon, ubounds=None, max_iter=50, u_threshold=1e-3,
                 ls_max_iter=10, ls_discount=0.2, ls_cost_threshold=0.3):
        self.model = model
        self.cost = cost
        self.dt = dt
        self.H = int(horizon)
        self.nx = model.state_dim
        self.obs_dim = model.system.obs_dim
        self.nu = model.system.ctrl_dim
        self.ubounds = ubounds
        self.max_iter = max_iter
        self.u_threshold = u_threshold
        self.ls_max_iter = ls_max_iter
        self.ls_discount = ls_discount
        self.ls_cost_threshold = ls_cost_threshold
        self.trace = []

    def _objective(self, xs, us):
        no = self.obs_dim
        obj = 0
        for i in range(self.H):
            obj += self.dt * (self.cost.eval_obs_cost(xs[i, :no]) + self.cost.eval_ctrl_cost(us[i]))
        obj += self.cost.eval_term_obs_cost(xs[-1, :no])
        return obj

    def _backward(self, states, ctrls, Jacs):
        nx, nu, no, H, dt = self.nx, self.nu, self.obs_dim, self.H, self.dt
        Ks = np.zeros((H, nu, nx))
        ks = np.zeros((H, nu))
        _, tj, th = self.cost.eval_term_obs_cost_hess(states[H, :no])
        Vn = np.zeros((nx, nx))
        vn = np.zeros(nx)
        Vn[:no, :no] = th
        vn[:no] = tj
        lin = quad = 0
        Ct = np.zeros((nx + nu, nx + nu))
        ct = np.zeros(nx + nu)
        for t in range(H, 0, -1):
            Qh = np.zeros((nx, nx))
            qx = np.zeros(nx)
            _, qx[:no], Qh[:no, :no] = self.cost.eval_obs_cost_hess(states[t - 1, :no])
            _, ru, Rh = self.cost.eval_ctrl_cost_hess(ctrls[t - 1])
            Ct[:nx, :nx] = Qh * dt
            Ct[nx:, nx:] = Rh * dt
            ct[:nx] = qx * dt
            ct[nx:] = ru * dt
            J = Jacs[t - 1]
            Qt = Ct + J.T @ Vn @ J
            qt = ct + J.T @ vn
            Quu, Qux, Qxu = Qt[nx:, nx:], Qt[nx:, :nx], Qt[:nx, nx:]
            K = -np.linalg.solve(Quu, Qux)
            k = -np.linalg.solve(Quu, qt[nx:])
            Ks[t - 1], ks[t - 1] = K, k
            lin += qt[nx:].dot(k)
            quad += k @ Quu @ k
            Vn = Qt[:nx, :nx] + Qxu @ K + K.T @ Qux + K.T @ Quu @ K
            vn = qt[:nx] + Qxu @ k + K.T @ (qt[nx:] + Quu @ k)
        return Ks, ks, lin, quad

    def _line_search_rollout(self, x0, states, ctrls, Ks, ks, alphas):
        L, H = len(alphas), self.H
        ls_states = np.zeros((L, H + 1, self.nx))
        ls_ctrls = np.zeros((L, H, self.nu))
        ls_states[:, 0, :] = x0
        for i in range(H):
            for j, alpha in enumerate(alphas):
                u = alpha * ks[i] + ctrls[i] + Ks[i] @ (ls_states[j, i, :] - states[i, :])
                if self.ubounds is not None:
                    u = np.clip(u, self.ubounds[0], self.ubounds[1])
                ls_ctrls[j, i, :] = u
            ls_states[:, i + 1, :] = self.model.pred_batch(ls_states[:, i, :], ls_ctrls[:, i, :])
        return ls_states, ls_ctrls

    def solve(self, x0, uguess):
        H, nx, nu = self.H, self.nx, self.nu
        self.trace = []
        states = np.zeros((H + 1, nx))
        ctrls = np.array(uguess, dtype=np.float64).reshape(H, nu).copy()
        Jacs = np.zeros((H, nx, nx + nu))
        states[0] = x0
        for i in range(H):
            states[i + 1], jx, ju = self.model.pred_diff(states[i], ctrls[i])
            Jacs[i, :, :nx] = jx
            Jacs[i, :, nx:] = ju
        obj = self._objective(states, ctrls)
        alphas = np.array([self.ls_discount ** i for i in range(self.ls_max_iter)])
        converged = False
        Ks = np.zeros((H, nu, nx))
        ks = np.zeros((H, nu))
        n_iter = 0
        for itr in range(self.max_iter):
            n_iter = itr + 1
            Ks, ks, lin, quad = self._backward(states, ctrls, Jacs)
            ks_norm = np.linalg.norm(ks)
            ls_states, ls_ctrls = self._line_search_rollout(x0, states, ctrls, Ks, ks, alphas)
            best_obj, best_idx = np.inf, None
            last = 0
            for lsitr, alpha in enumerate(alphas):
                last = lsitr
                new_obj = self._objective(ls_states[lsitr], ls_ctrls[lsitr])
                expect = alpha * lin + alpha ** 2 * quad / 2
                with np.errstate(divide="ignore", invalid="ignore"):
                    ratio = np.float64(obj - new_obj) / np.float64(-expect)
                if ratio > self.ls_cost_threshold:
                    best_obj, best_idx = new_obj, lsitr
                    break
                if new_obj < best_obj:
                    best_obj, best_idx = new_obj, lsitr
                if ks_norm < self.u_threshold:
                    break
            ls_success = False
            new_states, new_ctrls = ls_states[last], ls_ctrls[last]
            if best_obj < obj or ks_norm < self.u_threshold:
                if best_idx is None:
                    raise UnboundLocalError("best_alpha_idx referenced before assignment")
                ls_success = True
                new_states, new_ctrls = ls_states[best_idx], ls_ctrls[best_idx]
                _, jxs, jus = self.model.pred_diff_batch(new_states[:-1, :], new_ctrls)
                Jacs[:, :, :nx] = jxs
                Jacs[:, :, nx:] = jus
                new_obj = self._objective(new_states, new_ctrls)
            self.trace.append((float(obj), float(new_obj), -1 if best_idx is None else int(best_idx),
                               bool(ls_success), int(last)))
            if (not ls_success and new_obj > obj + 1e-3) or best_idx is None:
                break
            du_norm = np.linalg.norm(new_ctrls - ctrls)
            if du_norm < self.u_threshold:
                converged = True
            states = np.copy(new_states)
            ctrls = np.copy(new_ctrls)
            obj = new_obj
            if converged:
                break
        self.n_iter = n_iter
        self.final_obj = float(obj)
        return converged, states, ctrls, Ks, ks

    def run(self, constate, new_obs):
        state = self.model.update_state(constate[:-self.nu], constate[-self.nu:], new_obs)
        converged, states, ctrls, Ks, ks = self.solve(state, np.zeros((self.H, self.nu)))
        self.last = (converged, states, ctrls, Ks, ks)
        u = ctrls[0] + Ks[0] @ (state - states[0])
        return u, np.concatenate([state, u])

    def reset(self):
        """IterativeLQR.reset (ilqr.py:78-82): nothing the default solve reads survives a reset
        (every run() re-solves from the zero guess)."""
        self.last = None

    def traj_to_state(self, traj):
        return self.model.traj_to_state(traj)
