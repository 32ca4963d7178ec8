"""SINDy dynamics inference -- numpy restatement (oracle, test-only).  PARITY UNPINNED.

The reference's SINDy model (autompc/sysid/sindy.py:96-244, basis functions in
autompc/sysid/basis_funcs.py:8-126) delegates both fitting and prediction to the third-party
package ``pysindy~=1.0`` (requirements.txt:6), which is neither vendored under /root/reference
nor installed in the image, and the reference's tests that touch SINDy
(tests/test_pipeline.py:96-160) assert only hyper-parameter plumbing.  So there is no golden
vector and no reference run to pin this file against; it restates pysindy 1.0's published
semantics as the reference uses them, anchored on the reference's call sites:

  library   sindy.py:134-152: [identity] + for f in 1..trig_freq: [sin(f .), cos(f .)] (+ the four
            interaction terms x*sin(f y), y*sin(f x), x*cos(f y), y*cos(f x) when
            trig_interaction) + for d in 2..poly_degree: [v**d] (+ cross terms, not restated)
  features  pysindy CustomLibrary: for every library function (in list order), for every
            itertools.combinations(range(n_vars), n_args) (in order) one feature; variables are
            v = [x, u] (SINDy.predict concatenates controls after states)
  predict   sindy.py:173-179: discrete  x' = Theta(v) Xi' ;  continuous  x' = x + dt Theta(v) Xi'
  Jacobian  sindy.py:189-244, including its two quirks: the polynomial gradient omits the factor
            ``degree`` (basis_funcs.py:24-25), and every interaction feature is found twice by the
            name lookup (once through each argument order), so its gradient is counted twice.
"""
import itertools

import numpy as np


def build_library(n_vars, trig_freq=0, trig_interaction=False, poly_degree=1):
    """List of features: (kind, variable indices, parameter)."""
    funcs = [("id", 1, None)]
    for f in range(1, trig_freq + 1):
        funcs += [("sin", 1, f), ("cos", 1, f)]
        if trig_interaction:
            funcs += [("xsin", 2, f), ("xsin2", 2, f), ("xcos", 2, f), ("xcos2", 2, f)]
    for d in range(2, poly_degree + 1):
        funcs.append(("pow", 1, d))
    feats = []
    for kind, n_args, par in funcs:
        for c in itertools.combinations(range(n_vars), n_args):
            feats.append((kind, c, par))
    return feats


def eval_features(feats, V):
    cols = []
    for kind, c, par in feats:
        a = V[:, c[0]]
        b = V[:, c[1]] if len(c) > 1 else None
        if kind == "id":
            cols.append(a)
        elif kind == "sin":
            cols.append(np.sin(par * a))
        elif kind == "cos":
            cols.append(np.cos(par * a))
        elif kind == "xsin":          # x * sin(f y), (x, y) = (a, b)
            cols.append(a * np.sin(par * b))
        elif kind == "xsin2":         # second argument order: y * sin(f x)
            cols.append(b * np.sin(par * a))
        elif kind == "xcos":
            cols.append(a * np.cos(par * b))
        elif kind == "xcos2":
            cols.append(b * np.cos(par * a))
        elif kind == "pow":
            cols.append(a ** par)
        else:
            raise NotImplementedError(kind)
    return np.stack(cols, axis=1)


def feature_grads(feats, V, strict_reference=True):
    """d Theta / d v  as [m, n_feat, n_vars] with the reference's quirks when strict."""
    m, n_vars = V.shape
    G = np.zeros((m, len(feats), n_vars))
    twice = 2.0 if strict_reference else 1.0
    for k, (kind, c, par) in enumerate(feats):
        a = V[:, c[0]]
        b = V[:, c[1]] if len(c) > 1 else None
        if kind == "id":
            G[:, k, c[0]] = 1.0
        elif kind == "sin":
            G[:, k, c[0]] = par * np.cos(par * a)
        elif kind == "cos":
            G[:, k, c[0]] = -par * np.sin(par * a)
        elif kind == "xsin":
            G[:, k, c[0]] = twice * np.sin(par * b)
            G[:, k, c[1]] = twice * a * par * np.cos(par * b)
        elif kind == "xsin2":
            G[:, k, c[1]] = twice * np.sin(par * a)
            G[:, k, c[0]] = twice * b * par * np.cos(par * a)
        elif kind == "xcos":
            G[:, k, c[0]] = twice * np.cos(par * b)
            G[:, k, c[1]] = twice * a * -par * np.sin(par * b)
        elif kind == "xcos2":
            G[:, k, c[1]] = twice * np.cos(par * a)
            G[:, k, c[0]] = twice * b * -par * np.sin(par * a)
        elif kind == "pow":
            G[:, k, c[0]] = (1.0 if strict_reference else par) * a ** (par - 1)
    return G


class SINDyOracle:
    """Model-shaped SINDy surrogate with given coefficients Xi [nx, n_feat]."""

    def __init__(self, system, coefficients, trig_freq=0, trig_interaction=False, poly_degree=1,
                 time_mode="discrete", strict_reference=True):
        self.system = system
        nx, nu = system.obs_dim, system.ctrl_dim
        self.feats = build_library(nx + nu, trig_freq, trig_interaction, poly_degree)
        self.Xi = np.asarray(coefficients, dtype=np.float64).reshape(nx, len(self.feats))
        self.time_mode = time_mode
        self.strict_reference = strict_reference

    @property
    def state_dim(self):
        return self.system.obs_dim

    def traj_to_state(self, traj):
        return traj[-1].obs.copy()

    def update_state(self, state, new_ctrl, new_obs):
        return np.array(new_obs, dtype=np.float64)

    def pred_batch(self, states, ctrls):
        V = np.concatenate([states, ctrls], axis=1)
        y = eval_features(self.feats, V) @ self.Xi.T
        return y if self.time_mode == "discrete" else states + self.system.dt * y

    def pred(self, state, ctrl):
        return self.pred_batch(state[None, :], ctrl[None, :])[0]

    def pred_diff_batch(self, states, ctrls):
        nx = states.shape[1]
        V = np.concatenate([states, ctrls], axis=1)
        J = np.einsum("if,mfv->miv", self.Xi, feature_grads(self.feats, V, self.strict_reference))
        jx, ju = J[:, :, :nx].copy(), J[:, :, nx:].copy()
        if self.time_mode == "continuous":
            jx = np.eye(nx)[None] + self.system.dt * jx
            ju = self.system.dt * ju
        return self.pred_batch(states, ctrls), jx, ju

    def pred_diff(self, state, ctrl):
        o, a, b = self.pred_diff_batch(state[None, :], ctrl[None, :])
        return o[0], a[0], b[0]
