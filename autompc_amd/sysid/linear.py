"""Linear system-ID models (ARX, Koopman) whose prediction runs in HIP on MI355X.

Drop-ins for the reference's ``autompc.sysid.ARX`` (autompc/sysid/arx.py:42-187) and
``autompc.sysid.Koopman`` (autompc/sysid/koopman.py:82-196).  Both predict

    x' = A x + B u

on a model state that is richer than the observation: ARX stacks the last ``history``
observations and controls plus a constant 1 (arx.py:47-60), Koopman lifts the observation
through basis functions (koopman.py:112-122).  Fitting (least squares / lasso) is a one-off
host computation; ``pred`` / ``pred_batch`` / ``pred_diff`` / ``pred_diff_batch`` and every MPPI /
iLQR solve built on them go through the C ABI (``ampc_set_linear``), where the pair (A, B) is
staged as a one-hidden-layer identity-activation network so that the MFMA rollout, Jacobian and
iLQR kernels serve it unchanged.  Model states of up to 32 entries use two 16-column output tiles;
33..64 entries (long histories, large lifts) use the four-tile variant of the same kernels (hidden
width -- here the padded model state -- of at most 64).  65..256 entries (ARX with history 3..10 on
HalfCheetah is 66..235 states) run on a dedicated K-tiled MFMA kernel family
(csrc/linear_kernels.hpp): prediction, Jacobians, MPPI and the closed loop; iLQR is refused there.
"""
import numpy as np

from .. import _lib
from .model import Model, ModelFactory


def _config_space():
    try:
        import ConfigSpace as CS
        import ConfigSpace.conditions as CSC
        import ConfigSpace.hyperparameters as CSH
    except ImportError as e:           # ConfigSpace is an optional, tuner-side dependency
        raise ImportError("ConfigSpace is required for get_configuration_space()") from e
    return CS.ConfigurationSpace(), CSH, CSC


class _LinearModel(Model):
    """Shared device plumbing: A [ns, ns], B [ns, nu] are set by train() / set_parameters()."""

    def __init__(self, system, precision="f64", device=0):
        super().__init__(system)
        self.precision, self.device = precision, device
        self.A = self.B = None
        self._handle = None

    def stage_into(self, handle):
        if self.A is None:
            raise RuntimeError("%s is not trained" % type(self).__name__)
        handle.set_linear(self.A, self.B)

    def _dev(self):
        if self._handle is None:
            self._handle = _lib.Handle(self.device, self.precision)
            self.stage_into(self._handle)
        return self._handle

    def _invalidate(self):
        if self._handle is not None:
            self._handle.close()
        self._handle = None

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handle"] = None
        return state

    def pred(self, state, ctrl):
        return self._dev().pred_batch(np.asarray(state)[None, :], np.asarray(ctrl)[None, :])[0]

    def pred_batch(self, states, ctrls):
        return self._dev().pred_batch(states, ctrls)

    def pred_diff(self, state, ctrl):
        o, a, b = self._dev().pred_diff_batch(np.asarray(state)[None, :], np.asarray(ctrl)[None, :])
        return o[0], a[0], b[0]

    def pred_diff_batch(self, states, ctrls):
        return self._dev().pred_diff_batch(states, ctrls)

    def to_linear(self):
        return np.copy(self.A), np.copy(self.B)


# ------------------------------------------------------------------------------------- ARX
class ARX(_LinearModel):
    """state = [obs_t, (obs_{t-1}, ctrl_{t-1}), .., (obs_{t-k+1}, ctrl_{t-k+1}), 1]
    (arx.py:47-60, 104-105); the control of the current step is the model input."""

    # update_state(state, u, pred(state, u)[:obs_dim]) == pred(state, u): a closed loop on this model
    # may carry the predicted state forward (the device-resident evaluator does); Koopman re-lifts
    # the observation every step (koopman.py:160-168) and may not
    device_closed_loop = True

    def __init__(self, system, history=4, precision="f64", device=0):
        super().__init__(system, precision, device)
        self.k = int(history)
        if self.k < 1:
            raise ValueError("history must be >= 1")
        self.coeffs = None

    def _get_fvec_size(self):
        return 1 + self.k * self.system.obs_dim + self.k * self.system.ctrl_dim

    @property
    def state_dim(self):
        return self._get_fvec_size() - self.system.ctrl_dim

    def _get_all_feature_vectors(self, traj):
        """Row t = feature vector that predicts obs[t+1] (arx.py:62-76): the row's own obs,
        then lag-i obs/ctrl pairs padded with row 0, the constant, the row's own control."""
        no, nu, k = self.system.obs_dim, self.system.ctrl_dim, self.k
        obs, ctrls = np.asarray(traj.obs), np.asarray(traj.ctrls)
        T = obs.shape[0]
        F = np.zeros((T, self._get_fvec_size()))
        F[:, :no] = obs
        j = no
        for i in range(1, k):
            idx = np.maximum(np.arange(T) - i, 0)
            F[:, j:j + no] = obs[idx]
            j += no
            F[:, j:j + nu] = ctrls[idx]
            j += nu
        F[:, -(nu + 1)] = 1.0
        F[:, -nu:] = ctrls
        return F

    def _get_feature_vector(self, traj, t=None):
        """Feature vector built from rows < t (arx.py:47-60)."""
        if t is None:
            t = len(traj)
        return self._get_all_feature_vectors(traj[:t] if t < len(traj) else traj)[t - 1]

    def traj_to_state(self, traj):
        return self._get_feature_vector(traj)[:-self.system.ctrl_dim]

    def traj_to_states(self, traj):
        return self._get_all_feature_vectors(traj)[:, :-self.system.ctrl_dim]

    def state_to_obs(self, state):
        return state[0:self.system.obs_dim]

    # update_state shifts the OLD state (arx.py:113-127): a controller must hand it the state it was
    # given, whole (control/ilqr.py: traj_to_state)
    update_state_reads_state = True

    def update_state(self, state, new_ctrl, new_obs):
        # shift the history with the model itself, then overwrite the prediction with the
        # measured observation (arx.py:94-99)
        newstate = self.A @ np.asarray(state) + self.B @ np.asarray(new_ctrl)
        newstate[:self.system.obs_dim] = new_obs
        return newstate

    def _build_system_matrices(self, coeffs):
        """(A, B) from the regression coefficients [no, fvec] (arx.py:121-148)."""
        n, l, k = self.system.obs_dim, self.system.ctrl_dim, self.k
        m = n + l
        ns = self.state_dim
        A, B = np.zeros((ns, ns)), np.zeros((ns, l))
        A[-1, -1] = 1.0                                      # constant term
        if k > 1:
            A[n:2 * n, 0:n] = np.eye(n)                      # obs_t -> lag-1 slot
            B[2 * n:2 * n + l, :] = np.eye(l)                # u_t -> lag-1 control slot
        for i in range(k - 2):                               # lag-i block -> lag-(i+1) block
            A[(i + 1) * m + n:(i + 2) * m + n, i * m + n:(i + 1) * m + n] = np.eye(m)
        A[0:n, :] = coeffs[:, :-l]
        B[0:n, :] = coeffs[:, -l:]
        return A, B

    def train(self, trajs, silent=False):
        rows, targets = [], []
        for traj in trajs:                                   # arx.py:82-92: rows t = 1..len-1
            F = self._get_all_feature_vectors(traj)
            rows.append(F[:-1])
            targets.append(np.asarray(traj.obs)[1:])
        matrix, targets = np.concatenate(rows), np.concatenate(targets)
        coeffs = np.zeros((self.system.obs_dim, self._get_fvec_size()))
        for i in range(targets.shape[1]):
            coeffs[i, :] = np.linalg.lstsq(matrix, targets[:, i], rcond=None)[0]
        self._set_coeffs(coeffs)

    def _set_coeffs(self, coeffs):
        self.coeffs = np.array(coeffs, dtype=np.float64)
        self.A, self.B = self._build_system_matrices(self.coeffs)
        self._invalidate()

    def get_parameters(self):
        return {"coeffs": np.copy(self.coeffs)}

    def set_parameters(self, params):
        self._set_coeffs(params["coeffs"])


class ARXFactory(ModelFactory):
    """Hyper-parameter: history, int 1..10, default 4 (arx.py:33-40)."""
    Model = ARX
    name = "ARX"

    def get_configuration_space(self):
        cs, CSH, _ = _config_space()
        cs.add_hyperparameter(CSH.UniformIntegerHyperparameter("history", lower=1, upper=10,
                                                               default_value=4))
        return cs


# --------------------------------------------------------------------------------- Koopman
def _as_bool(v):
    return (v == "true") if isinstance(v, str) else bool(v)


class Koopman(_LinearModel):
    """state = basis functions applied to the observation, basis-major:
    [f0(o_0..o_n), f1(o_0..o_n), ..] with f0 the identity (koopman.py:105-122).

    strict_reference=True (default) reproduces which functions the reference actually ends up
    with: its lambdas are created in loops and all see the loop variable's FINAL value
    (koopman.py:107-110), so every polynomial term is ``x**poly_degree`` and every trig pair is
    ``sin/cos(poly_degree * x)`` (the loop also runs to ``poly_degree``, not ``trig_freq``).
    strict_reference=False gives the documented basis: x**2..x**poly_degree and
    sin/cos(i x), i = 1..trig_freq."""

    def __init__(self, system, method="lstsq", lasso_alpha=None, poly_basis=False, poly_degree=1,
                 trig_basis=False, trig_freq=1, product_terms=False, use_cuda=None,
                 strict_reference=True, precision="f64", device=0):
        super().__init__(system, precision, device)
        if method not in ("lstsq", "lasso", "stable"):
            raise ValueError("method must be lstsq, lasso or stable")
        self.method = method
        self.lasso_alpha = lasso_alpha
        self.poly_basis, self.trig_basis = _as_bool(poly_basis), _as_bool(trig_basis)
        self.poly_degree, self.trig_freq = int(poly_degree), int(trig_freq)
        self.product_terms = _as_bool(product_terms)
        self.strict_reference = bool(strict_reference)
        # (kind, parameter): 0 identity, 1 power, 2 sin, 3 cos
        basis = [(0, 1)]
        if self.poly_basis:
            for i in range(2, 1 + self.poly_degree):
                basis.append((1, self.poly_degree if self.strict_reference else i))
        if self.trig_basis:
            last = self.poly_degree if self.strict_reference else self.trig_freq
            for i in range(1, 1 + last):
                f = last if self.strict_reference else i
                basis += [(2, f), (3, f)]
        self.basis = basis

    def _apply_basis(self, obs):
        obs = np.asarray(obs, dtype=np.float64)
        parts = []
        for kind, p in self.basis:
            parts.append(obs if kind == 0 else obs ** p if kind == 1
                         else np.sin(p * obs) if kind == 2 else np.cos(p * obs))
        lifted = np.concatenate(parts, axis=-1)
        if self.product_terms:                               # koopman.py:114-120
            n = lifted.shape[-1]
            iu = np.triu_indices(n, k=1)
            lifted = np.concatenate([lifted, lifted[..., iu[0]] * lifted[..., iu[1]]], axis=-1)
        return lifted

    def device_lift(self):
        """(kinds, params) of the basis for the device closed loop (ampc_mppi_plan_set_state_lift), or
        None when the lift has product terms (not expressible there)."""
        if self.product_terms:
            return None
        return (np.array([k for k, _ in self.basis], dtype=np.int32),
                np.array([p for _, p in self.basis], dtype=np.float64))

    def _transform_observations(self, observations):
        return self._apply_basis(np.asarray(observations))

    def traj_to_state(self, traj):
        return self._apply_basis(traj.obs[-1])

    def traj_to_states(self, traj):
        return self._transform_observations(traj.obs[:])

    def update_state(self, state, new_ctrl, new_obs):
        return self._apply_basis(new_obs)

    @property
    def state_dim(self):
        n = len(self.basis) * self.system.obs_dim
        if self.product_terms and not self.strict_reference:
            n += n * (n - 1) // 2          # the reference forgets these (koopman.py:138-139)
        return n

    def train(self, trajs, silent=False):
        if self.method == "stable":
            raise NotImplementedError("method='stable' (stable_koopman.py) is a training-time "
                                      "optimisation outside the MPC path; fit with lstsq or lasso")
        lifted = [self._transform_observations(t.obs[:]) for t in trajs]
        X = np.concatenate([z[:-1] for z in lifted]).T
        Y = np.concatenate([z[1:] for z in lifted]).T
        U = np.concatenate([np.asarray(t.ctrls)[:-1] for t in trajs]).T
        n = X.shape[0]
        XU = np.concatenate([X, U], axis=0)
        if self.method == "lstsq":                           # koopman.py:151-154
            AB = Y @ np.linalg.pinv(XU)
        else:                                                # lasso, koopman.py:155-161
            from sklearn.linear_model import Lasso
            clf = Lasso(alpha=self.lasso_alpha)
            clf.fit(XU.T, Y.T)
            AB = np.atleast_2d(clf.coef_)
        self._set_matrices(AB[:n, :n], AB[:n, n:])

    def _set_matrices(self, A, B):
        self.A, self.B = np.array(A, dtype=np.float64), np.array(B, dtype=np.float64)
        self._invalidate()

    def get_parameters(self):
        return {"A": np.copy(self.A), "B": np.copy(self.B)}

    def set_parameters(self, params):
        self._set_matrices(params["A"], params["B"])


class KoopmanFactory(ModelFactory):
    """Hyper-parameters as koopman.py:47-80: method {lstsq, lasso, stable}; lasso_alpha log
    1e-10..1e2 (method == lasso); poly_basis / poly_degree 2..8; trig_basis / trig_freq 1..8;
    product_terms {false}."""
    Model = Koopman
    name = "Koopman"

    def get_configuration_space(self):
        cs, CSH, CSC = _config_space()
        method = CSH.CategoricalHyperparameter("method", choices=["lstsq", "lasso", "stable"])
        alpha = CSH.UniformFloatHyperparameter("lasso_alpha", lower=1e-10, upper=1e2,
                                               default_value=1.0, log=True)
        poly = CSH.CategoricalHyperparameter("poly_basis", choices=["true", "false"],
                                             default_value="false")
        degree = CSH.UniformIntegerHyperparameter("poly_degree", lower=2, upper=8, default_value=3)
        trig = CSH.CategoricalHyperparameter("trig_basis", choices=["true", "false"],
                                             default_value="false")
        freq = CSH.UniformIntegerHyperparameter("trig_freq", lower=1, upper=8, default_value=1)
        prod = CSH.CategoricalHyperparameter("product_terms", choices=["false"],
                                             default_value="false")
        cs.add_hyperparameters([method, poly, degree, trig, freq, prod, alpha])
        cs.add_conditions([CSC.InCondition(child=degree, parent=poly, values=["true"]),
                           CSC.InCondition(child=freq, parent=trig, values=["true"]),
                           CSC.InCondition(child=alpha, parent=method, values=["lasso"])])
        return cs
