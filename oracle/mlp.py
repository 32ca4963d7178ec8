"""MLP dynamics step and its Jacobian -- numpy restatement (oracle, test-only).

Follows the reference's arithmetic order:
  normalise   autompc/sysid/mlp.py:20-24   (x - mean) / std, column by column
  network     autompc/sysid/mlp.py:55-59   act(W x + b) per hidden layer, linear out
  denormalise autompc/sysid/mlp.py:26-30   y * dy_std + dy_mean
  residual    autompc/sysid/mlp.py:236     state + dy
  Jacobian    autompc/sysid/mlp.py:288-305 d(net)/d(input) / xu_std * dy_std[:,None], +I on
              the state block.  The reference gets d(net)/d(input) from autograd;
              here it is the analytic chain  W_out * prod_l diag(act'(z_l)) W_l.
"""
import numpy as np

SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946

ACTIVATIONS = ("relu", "tanh", "sigmoid", "selu")


def act_fn(name, z):
    if name == "relu":
        return np.maximum(z, 0.0)
    if name == "tanh":
        return np.tanh(z)
    if name == "sigmoid":
        return 1.0 / (1.0 + np.exp(-z))
    if name == "selu":
        return SELU_SCALE * np.where(z > 0, z, SELU_ALPHA * np.expm1(np.minimum(z, 0.0)))
    raise NotImplementedError(name)


def act_grad(name, z):
    if name == "relu":
        return (z > 0).astype(z.dtype)
    if name == "tanh":
        t = np.tanh(z)
        return 1.0 - t * t
    if name == "sigmoid":
        s = 1.0 / (1.0 + np.exp(-z))
        return s * (1.0 - s)
    if name == "selu":
        return SELU_SCALE * np.where(z > 0, 1.0, SELU_ALPHA * np.exp(np.minimum(z, 0.0)))
    raise NotImplementedError(name)


def make_params(weights, biases, activation, xu_means, xu_std, dy_means, dy_std):
    """weights[l] has torch.nn.Linear layout [out_l, in_l]; the last entry is the
    linear output layer."""
    return {
        "weights": [np.asarray(w, dtype=np.float64) for w in weights],
        "biases": [np.asarray(b, dtype=np.float64) for b in biases],
        "activation": activation,
        "xu_means": np.asarray(xu_means, dtype=np.float64),
        "xu_std": np.asarray(xu_std, dtype=np.float64),
        "dy_means": np.asarray(dy_means, dtype=np.float64),
        "dy_std": np.asarray(dy_std, dtype=np.float64),
    }


def random_params(nx, nu, hidden, activation="relu", seed=0, dy_std=0.1):
    """Synthetic weights with torch.nn.Linear's default init distribution
    (U(-1/sqrt(in), 1/sqrt(in)) for W and b), drawn from numpy so that the GPU
    box needs no reference.  Normalisers: mean 0 / std 1 in, mean 0 / std
    ``dy_std`` out (SURVEY.md 8d synthetic-input recipe)."""
    rng = np.random.default_rng(seed)
    sizes = [nx + nu] + list(hidden) + [nx]
    Ws, bs = [], []
    for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
        bound = 1.0 / np.sqrt(fan_in)
        Ws.append(rng.uniform(-bound, bound, size=(fan_out, fan_in)))
        bs.append(rng.uniform(-bound, bound, size=(fan_out,)))
    return make_params(Ws, bs, activation, np.zeros(nx + nu), np.ones(nx + nu),
                       np.zeros(nx), np.full(nx, dy_std))


def _net(params, Xt, want_grads=False):
    name = params["activation"]
    h = Xt
    grads = []
    for W, b in zip(params["weights"][:-1], params["biases"][:-1]):
        z = h @ W.T + b
        if want_grads:
            grads.append(act_grad(name, z))
        h = act_fn(name, z)
    y = h @ params["weights"][-1].T + params["biases"][-1]
    return (y, grads) if want_grads else y


def pred_batch(params, states, ctrls):
    X = np.concatenate([states, ctrls], axis=1)
    Xt = (X - params["xu_means"]) / params["xu_std"]
    y = _net(params, Xt)
    return states + (y * params["dy_std"] + params["dy_means"])


def pred_diff_batch(params, states, ctrls):
    nx = states.shape[1]
    X = np.concatenate([states, ctrls], axis=1)
    Xt = (X - params["xu_means"]) / params["xu_std"]
    y, grads = _net(params, Xt, want_grads=True)
    Ws = params["weights"]
    # J_net[m] = W_out diag(g_L) W_L ... diag(g_1) W_1, built left to right.
    J = np.broadcast_to(Ws[-1], (X.shape[0],) + Ws[-1].shape)
    for W, g in zip(reversed(Ws[:-1]), reversed(grads)):
        J = (J * g[:, None, :]) @ W
    J = J / params["xu_std"][None, None, :] * params["dy_std"][None, :, None]
    state_jac = J[:, :, :nx] + np.eye(nx)[None]
    ctrl_jac = J[:, :, nx:].copy()
    out = states + (y * params["dy_std"] + params["dy_means"])
    return out, state_jac, ctrl_jac


class MLPOracle:
    """Model-shaped wrapper (reference Model surface, autompc/sysid/model.py:55-244)."""

    def __init__(self, system, params):
        self.system = system
        self.params = params

    @property
    def state_dim(self):
        return self.system.obs_dim

    def traj_to_state(self, traj):
        return traj[-1].obs.copy()

    def update_state(self, state, new_ctrl, new_obs):
        return np.array(new_obs, dtype=np.float64)

    def pred(self, state, ctrl):
        return pred_batch(self.params, state[None, :], ctrl[None, :])[0]

    def pred_batch(self, states, ctrls):
        return pred_batch(self.params, states, ctrls)

    def pred_diff(self, state, ctrl):
        o, a, b = pred_diff_batch(self.params, state[None, :], ctrl[None, :])
        return o[0], a[0], b[0]

    def pred_diff_batch(self, states, ctrls):
        return pred_diff_batch(self.params, states, ctrls)


class MLPOracleTorch(MLPOracle):
    """The same model with the reference's own CALL STRUCTURE (for the CPU baseline's timing, SURVEY.md
    section 8d): torch f64 ``nn.Linear`` layers on the CPU, a numpy -> torch -> numpy copy per call and
    the column-by-column Python normalisation loops (mlp.py:20-30 transform_input / transform_output,
    :55-59 ForwardNet.forward, :219-236 pred / pred_batch), Jacobians by autograd over an obs_dim-fold
    repeated batch (mlp.py:238-305).  Same values as the numpy restatement to rounding
    (tests/test_oracle_golden.py)."""

    def __init__(self, system, params):
        super().__init__(system, params)
        import torch
        self._torch = torch
        self._W = [torch.from_numpy(np.ascontiguousarray(w)) for w in params["weights"]]
        self._b = [torch.from_numpy(np.ascontiguousarray(b)) for b in params["biases"]]
        self._act = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "selu": torch.selu}[
            params["activation"]]

    @staticmethod
    def _transform_input(means, std, XU):                 # mlp.py:20-24
        cols = []
        for i in range(XU.shape[1]):
            cols.append((XU[:, i] - means[i]) / std[i])
        return np.vstack(cols).T

    @staticmethod
    def _transform_output(means, std, XU):                # mlp.py:26-30
        cols = []
        for i in range(XU.shape[1]):
            cols.append((XU[:, i] * std[i]) + means[i])
        return np.vstack(cols).T

    def _forward(self, x):                                # mlp.py:55-59
        F = self._torch.nn.functional
        for W, b in zip(self._W[:-1], self._b[:-1]):
            x = self._act(F.linear(x, W, b))
        return F.linear(x, self._W[-1], self._b[-1])

    def pred_batch(self, states, ctrls):                  # mlp.py:229-236
        p = self.params
        X = np.concatenate([states, ctrls], axis=1)
        Xt = self._transform_input(p["xu_means"], p["xu_std"], X)
        with self._torch.no_grad():
            yout = self._forward(self._torch.from_numpy(Xt)).cpu().numpy()
        dy = self._transform_output(p["dy_means"], p["dy_std"], yout).flatten()
        return states + dy.reshape((states.shape[0], self.state_dim))

    def pred(self, state, ctrl):                          # mlp.py:219-227
        return self.pred_batch(state[None, :], ctrl[None, :])[0]

    def pred_diff_batch(self, states, ctrls):             # mlp.py:281-305
        torch, p = self._torch, self.params
        X = np.concatenate([states, ctrls], axis=1)
        Xt = self._transform_input(p["xu_means"], p["xu_std"], X)
        n, m = states.shape[1], states.shape[0]
        T = torch.from_numpy(Xt).repeat(n, 1, 1).permute(1, 0, 2).flatten(0, 1)
        T.requires_grad_(True)
        predy = self._forward(T)
        predy.backward(torch.eye(n, dtype=predy.dtype).repeat(m, 1), retain_graph=True)
        predy = predy.reshape((m, n, n))
        jac = T.grad.cpu().data.numpy().reshape((m, n, T.shape[-1]))
        jac = jac / np.tile(p["xu_std"], (m, n, 1)) * np.tile(p["dy_std"], (m, 1))[:, :, np.newaxis]
        out = predy[:, 0, :].cpu().data.numpy()
        dy = self._transform_output(p["dy_means"], p["dy_std"], out)
        return states + dy, jac[:, :, :n] + np.tile(np.eye(n), (m, 1, 1)), jac[:, :, n:]

    def pred_diff(self, state, ctrl):                     # mlp.py:238-279 (same values as the batch form)
        o, a, b = self.pred_diff_batch(state[None, :], ctrl[None, :])
        return o[0], a[0], b[0]
