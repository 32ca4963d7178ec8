"""SINDy surrogate dynamics whose inference runs in HIP on MI355X.

Mirrors the reference's ``autompc.sysid.SINDy`` / ``SINDyFactory`` (reference:
autompc/sysid/sindy.py:24-253; basis functions autompc/sysid/basis_funcs.py:8-126): same
constructor hyper-parameters, same library (identity; sin / cos up to ``trig_freq``; the four trig
interaction terms; powers up to ``poly_degree``; the polynomial cross terms of
basis_funcs.py:27-93), same ``time_mode`` semantics

    discrete:    x' = Theta([x,u]) Xi'          continuous:    x' = x + dt Theta([x,u]) Xi'

Inference (prediction, the name-lookup Jacobian with its two quirks, MPPI and iLQR on the model)
is pinned by tests/golden/sindy_*.npz, generated from the reference's own code (see
oracle/sindy.py); only the feature ORDER rests on the documented enumeration of the absent
``pysindy~=1.0`` package.  ``train`` (out of the hot path) is a numpy sequentially-thresholded
least squares (what ``ps.STLSQ`` does: ridge alpha 0.05, 20 iterations) and is not pinned.
"""
import itertools

import numpy as np

from .. import _lib
from .model import Model, ModelFactory

K_ID, K_SIN, K_COS, K_XSIN, K_XCOS, K_POW, K_MONO = range(7)


def cross_term_exponents(degree):
    """The exponent tuples get_cross_term_basis_funcs(degree) turns into basis functions, in its
    order (basis_funcs.py:27-40): all exponent vectors in {0..degree-1}^degree (last axis fastest)
    that sum to `degree`, zeros dropped, first occurrence of every trimmed tuple kept.  Each
    tuple (e_0, .., e_{n-1}) is the n-argument function prod_j x_j ** e_j."""
    out, seen = [], set()
    for exp in itertools.product(range(degree), repeat=degree):
        if sum(exp) != degree:
            continue
        tr = tuple(e for e in exp if e > 0)
        if tr not in seen:
            seen.add(tr)
            out.append(tr)
    return out


def build_library(n_vars, trig_freq=0, trig_interaction=False, poly_degree=1, poly_cross_terms=False):
    """Feature descriptors (kind, a, b, param, pair_var, pair_exp) in the order pysindy's
    CustomLibrary enumerates them: library functions in list order, each over
    itertools.combinations of the variables.  A monomial feature (kind K_MONO, the polynomial cross
    terms) has a = index of its first (variable, exponent) pair in pair_var / pair_exp and b = its
    number of pairs."""
    kind, a0, a1, par = [], [], [], []
    pair_var, pair_exp = [], []

    def add(k, a, b, p):
        kind.append(k); a0.append(a); a1.append(b); par.append(float(p))
    for i in range(n_vars):
        add(K_ID, i, i, 0.0)
    for f in range(1, trig_freq + 1):
        for i in range(n_vars):
            add(K_SIN, i, i, f)
        for i in range(n_vars):
            add(K_COS, i, i, f)
        if trig_interaction:
            pairs = list(itertools.combinations(range(n_vars), 2))
            for a, b in pairs:
                add(K_XSIN, a, b, f)        # x sin(f y)
            for a, b in pairs:
                add(K_XSIN, b, a, f)        # second argument order: y sin(f x)
            for a, b in pairs:
                add(K_XCOS, a, b, f)
            for a, b in pairs:
                add(K_XCOS, b, a, f)
    for d in range(2, poly_degree + 1):
        for i in range(n_vars):
            add(K_POW, i, i, d)
    if poly_cross_terms:                     # sindy.py:143-145: after ALL the pure powers
        for d in range(2, poly_degree + 1):
            for tr in cross_term_exponents(d):
                if len(tr) > 10:
                    raise ValueError("n_args > 10")          # as basis_funcs.py:71-72
                for combo in itertools.combinations(range(n_vars), len(tr)):
                    add(K_MONO, len(pair_var), len(tr), 0.0)
                    pair_var.extend(combo)
                    pair_exp.extend(tr)
    return (np.array(kind, dtype=np.int32), np.array(a0, dtype=np.int32),
            np.array(a1, dtype=np.int32), np.array(par, dtype=np.float64),
            np.array(pair_var, dtype=np.int32), np.array(pair_exp, dtype=np.int32))


def _features(lib, V):
    kind, a0, a1, par, pair_var, pair_exp = lib
    mono = kind == K_MONO
    A = V[:, np.where(mono, 0, a0)]
    B = V[:, np.where(mono, 0, a1)]
    out = np.empty_like(A)
    for k in np.nonzero(mono)[0]:
        val = np.ones(V.shape[0])
        for j in range(a0[k], a0[k] + a1[k]):
            val = val * V[:, pair_var[j]] ** int(pair_exp[j])
        out[:, k] = val
    for k, fn in ((K_ID, lambda a, b, p: a), (K_SIN, lambda a, b, p: np.sin(p * a)),
                  (K_COS, lambda a, b, p: np.cos(p * a)), (K_XSIN, lambda a, b, p: a * np.sin(p * b)),
                  (K_XCOS, lambda a, b, p: a * np.cos(p * b)), (K_POW, lambda a, b, p: a ** p)):
        m = kind == k
        if m.any():
            out[:, m] = fn(A[:, m], B[:, m], par[m])
    return out


def _as_bool(v):
    return (v == "true") if isinstance(v, str) else bool(v)


class SINDy(Model):
    def __init__(self, system, method="lstsq", lasso_alpha=None, threshold=1e-2, poly_basis=False,
                 poly_degree=1, poly_cross_terms=False, trig_basis=False, trig_freq=1,
                 trig_interaction=False, time_mode="discrete", precision="f64", device=0,
                 strict_reference=True):
        super().__init__(system)
        if time_mode not in ("discrete", "continuous"):
            raise ValueError("time_mode must be 'discrete' or 'continuous'")
        self.method, self.lasso_alpha, self.threshold = method, lasso_alpha, threshold
        self.poly_basis, self.poly_degree = _as_bool(poly_basis), int(poly_degree)
        self.poly_cross_terms = _as_bool(poly_cross_terms)
        self.trig_basis, self.trig_freq = _as_bool(trig_basis), int(trig_freq)
        self.trig_interaction = _as_bool(trig_interaction)
        self.time_mode = time_mode
        self.precision, self.device, self.strict_reference = precision, device, strict_reference
        n = system.obs_dim + system.ctrl_dim
        self.library = build_library(n, self.trig_freq if self.trig_basis else 0,
                                     self.trig_basis and self.trig_interaction,
                                     self.poly_degree if self.poly_basis else 1,
                                     self.poly_basis and self.poly_cross_terms)
        self.coefficients = np.zeros((system.obs_dim, self.library[0].shape[0]))
        self._handle = None

    # -- reference Model surface (sindy.py:120-128) --------------------------------------
    def traj_to_state(self, traj):
        return traj[-1].obs.copy()

    def update_state(self, state, new_ctrl, new_obs):
        return np.array(new_obs, dtype=np.float64)

    @property
    def state_dim(self):
        return self.system.obs_dim

    # -- parameters ----------------------------------------------------------------------
    def set_coefficients(self, xi):
        xi = np.asarray(xi, dtype=np.float64)
        if xi.shape != self.coefficients.shape:
            raise ValueError("coefficients must have shape %r" % (self.coefficients.shape,))
        self.coefficients = xi.copy()
        self._invalidate()

    def get_parameters(self):
        return {"coefficients": self.coefficients.copy()}

    def set_parameters(self, params):
        self.set_coefficients(params["coefficients"])

    # -- fit: sequentially thresholded least squares (host, outside the hot path) --------
    def train(self, trajs, xdot=None, silent=False, alpha=0.05, max_iter=20):
        X = np.concatenate([t.obs[:-1] for t in trajs])
        U = np.concatenate([t.ctrls[:-1] for t in trajs])
        if self.time_mode == "discrete":
            Y = np.concatenate([t.obs[1:] for t in trajs])
        elif xdot is not None:
            Y = np.concatenate([np.asarray(d)[:-1] for d in xdot])
        else:
            Y = np.concatenate([np.gradient(t.obs, self.system.dt, axis=0)[:-1] for t in trajs])
        Theta = _features(self.library, np.concatenate([X, U], axis=1))
        nf = Theta.shape[1]
        Xi = np.zeros((Y.shape[1], nf))
        for i in range(Y.shape[1]):
            keep = np.ones(nf, dtype=bool)
            coef = np.zeros(nf)
            for _ in range(max_iter):
                if not keep.any():
                    break
                A = Theta[:, keep]
                w = np.linalg.solve(A.T @ A + alpha * np.eye(A.shape[1]), A.T @ Y[:, i])
                coef[:] = 0.0
                coef[keep] = w
                small = np.abs(coef) < self.threshold
                if not (small & keep).any():
                    break
                keep &= ~small
            Xi[i] = np.where(keep, coef, 0.0)
        self.set_coefficients(Xi)

    # -- device staging / inference (HIP) --------------------------------------------------
    def stage_into(self, handle):
        kind, a0, a1, par, pair_var, pair_exp = self.library
        handle.set_sindy(self.system.obs_dim, self.system.ctrl_dim, kind, a0, a1, par,
                         self.coefficients, self.time_mode == "continuous", self.system.dt,
                         self.strict_reference, pair_var=pair_var, pair_exp=pair_exp)

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


class SINDyFactory(ModelFactory):
    """Hyper-parameter space of the reference's SINDyFactory (sindy.py:57-94)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.Model = SINDy
        self.name = "SINDy"

    def get_configuration_space(self):
        try:
            import ConfigSpace as CS
            import ConfigSpace.conditions as CSC
            import ConfigSpace.hyperparameters as CSH
        except ImportError as e:
            raise ImportError("ConfigSpace is required for get_configuration_space()") from e
        cs = CS.ConfigurationSpace()
        tf = ["true", "false"]
        time_mode = CSH.CategoricalHyperparameter("time_mode", choices=["discrete", "continuous"])
        method = CSH.CategoricalHyperparameter("method", choices=["lstsq", "lasso"])
        threshold = CSH.UniformFloatHyperparameter("threshold", lower=1e-5, upper=1e1,
                                                   default_value=1e-2, log=True)
        lasso_alpha = CSH.UniformFloatHyperparameter("lasso_alpha", lower=1e-5, upper=1e2,
                                                     default_value=1.0, log=True)
        poly_basis = CSH.CategoricalHyperparameter("poly_basis", choices=tf, default_value="false")
        poly_degree = CSH.UniformIntegerHyperparameter("poly_degree", lower=2, upper=8, default_value=3)
        poly_cross = CSH.CategoricalHyperparameter("poly_cross_terms", choices=tf, default_value="false")
        trig_basis = CSH.CategoricalHyperparameter("trig_basis", choices=tf, default_value="false")
        trig_freq = CSH.UniformIntegerHyperparameter("trig_freq", lower=1, upper=8, default_value=1)
        trig_inter = CSH.CategoricalHyperparameter("trig_interaction", choices=tf, default_value="false")
        cs.add_hyperparameters([method, lasso_alpha, threshold, poly_basis, poly_degree, trig_basis,
                                trig_freq, trig_inter, poly_cross, time_mode])
        cs.add_conditions([CSC.InCondition(child=lasso_alpha, parent=method, values=["lasso"]),
                           CSC.InCondition(child=poly_degree, parent=poly_basis, values=["true"]),
                           CSC.InCondition(child=trig_freq, parent=trig_basis, values=["true"]),
                           CSC.InCondition(child=trig_inter, parent=trig_basis, values=["true"])])
        return cs
