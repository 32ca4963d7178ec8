"""Linear system-ID models -- numpy restatement (ORACLE: test infrastructure only; nothing in
the product path may import this).

reference: autompc/sysid/arx.py:42-187 (ARX), autompc/sysid/koopman.py:82-196 (Koopman).
Pinned to the reference by tests/golden/linear_*.npz (gen_golden.py gen_linear): fitted A/B,
state construction, prediction, and MPPI / iLQR solves on top.

Both predict x' = A x + B u (arx.py:151-164, koopman.py:170-184); they differ in what the model
state is and how (A, B) are fitted.
"""
import numpy as np


class LinearOracle:
    def __init__(self, system, A=None, B=None):
        self.system = system
        self.A = None if A is None else np.array(A, dtype=np.float64)
        self.B = None if B is None else np.array(B, dtype=np.float64)

    def pred(self, state, ctrl):
        return self.A @ state + self.B @ ctrl

    def pred_batch(self, states, ctrls):
        return (self.A @ states.T + self.B @ ctrls.T).T

    def pred_diff(self, state, ctrl):
        return self.A @ state + self.B @ ctrl, np.copy(self.A), np.copy(self.B)

    def pred_diff_batch(self, states, ctrls):
        # Model.pred_diff_batch default: a loop over pred_diff (model.py:176-184)
        m = states.shape[0]
        return (self.pred_batch(states, ctrls), np.tile(self.A, (m, 1, 1)), np.tile(self.B, (m, 1, 1)))


class ARXOracle(LinearOracle):
    def __init__(self, system, history, A=None, B=None):
        super().__init__(system, A, B)
        self.k = int(history)

    def fvec_size(self):
        return 1 + self.k * (self.system.obs_dim + self.system.ctrl_dim)

    @property
    def state_dim(self):
        return self.fvec_size() - self.system.ctrl_dim

    def feature_vector(self, obs, ctrls, t=None):
        """arx.py:47-60: newest observation, then lag pairs (padded with row 0), 1, newest ctrl."""
        if t is None:
            t = obs.shape[0]
        parts = [obs[t - 1]]
        for i in range(t - 2, t - self.k - 1, -1):
            j = i if i >= 0 else 0
            parts += [obs[j], ctrls[j]]
        parts += [np.ones(1), ctrls[t - 1]]
        return np.concatenate(parts)

    def traj_to_state(self, traj):
        return self.feature_vector(np.asarray(traj.obs), np.asarray(traj.ctrls))[:-self.system.ctrl_dim]

    def state_from_first_obs(self, obs):
        return self.feature_vector(np.asarray(obs)[None, :], np.zeros((1, self.system.ctrl_dim)))[
            :-self.system.ctrl_dim]

    def update_state(self, state, new_ctrl, new_obs):
        new = self.A @ state + self.B @ new_ctrl        # arx.py:94-99
        new[:self.system.obs_dim] = new_obs
        return new

    def train(self, obs_list, ctrl_list):
        rows, tgt = [], []
        for obs, ctrls in zip(obs_list, ctrl_list):     # arx.py:82-92
            for t in range(1, obs.shape[0]):
                rows.append(self.feature_vector(obs, ctrls, t))
                tgt.append(obs[t])
        M, Y = np.array(rows), np.array(tgt)
        n, l, k = self.system.obs_dim, self.system.ctrl_dim, self.k
        coeffs = np.array([np.linalg.lstsq(M, Y[:, i], rcond=None)[0] for i in range(n)])
        ns, m = self.state_dim, n + l
        A, B = np.zeros((ns, ns)), np.zeros((ns, l))    # arx.py:121-148
        A[-1, -1] = 1.0
        if k > 1:
            A[n:2 * n, 0:n] = np.eye(n)
        for i in range(k - 2):
            A[(i + 1) * m + n:(i + 2) * m + n, i * m + n:(i + 1) * m + n] = np.eye(m)
        A[0:n, :] = coeffs[:, :-l]
        B[0:n, :] = coeffs[:, -l:]
        B[2 * n:2 * n + l, :] = np.eye(l)
        self.A, self.B = A, B


class KoopmanOracle(LinearOracle):
    """Basis as the reference ENDS UP with (koopman.py:105-110): lambdas made in loops share the
    loop variable, so all polynomial terms are x**poly_degree and all trig terms use frequency
    poly_degree (the trig loop also runs to poly_degree, not trig_freq)."""

    def __init__(self, system, poly_basis=False, poly_degree=1, trig_basis=False, A=None, B=None):
        super().__init__(system, A, B)
        d = int(poly_degree)
        funcs = [lambda x: x]
        if poly_basis:
            funcs += [(lambda x: x ** d)] * len(range(2, 1 + d))
        if trig_basis:
            for _ in range(1, 1 + d):
                funcs += [lambda x: np.sin(d * x), lambda x: np.cos(d * x)]
        self.funcs = funcs

    @property
    def state_dim(self):
        return len(self.funcs) * self.system.obs_dim

    def apply_basis(self, obs):
        return np.array([f(x) for f in self.funcs for x in obs])     # koopman.py:112-113

    def traj_to_state(self, traj):
        return self.apply_basis(np.asarray(traj.obs)[-1])

    def state_from_first_obs(self, obs):
        return self.apply_basis(obs)

    def update_state(self, state, new_ctrl, new_obs):
        return self.apply_basis(new_obs)

    def train(self, obs_list, ctrl_list, method="lstsq", lasso_alpha=None):
        lifted = [np.array([self.apply_basis(o) for o in obs]) for obs in obs_list]
        X = np.concatenate([z[:-1] for z in lifted]).T
        Y = np.concatenate([z[1:] for z in lifted]).T
        U = np.concatenate([c[:-1] for c in ctrl_list]).T
        n = X.shape[0]
        XU = np.concatenate([X, U], axis=0)
        if method == "lstsq":
            AB = Y @ np.linalg.pinv(XU)                              # koopman.py:151-154
        else:
            from sklearn.linear_model import Lasso                   # koopman.py:155-161
            clf = Lasso(alpha=lasso_alpha)
            clf.fit(XU.T, Y.T)
            AB = clf.coef_
        self.A, self.B = AB[:n, :n], AB[:n, n:]
