"""MLP surrogate dynamics whose inference and Jacobians run in HIP on MI355X.

Drop-in for the reference's ``autompc.sysid.MLP`` (reference:
autompc/sysid/mlp.py:137-321): same constructor hyper-parameters, same
``get_parameters`` / ``set_parameters`` dictionary (``net_state`` uses the
reference's ``state_dict`` keys ``layers.layer{i}.weight|bias`` /
``output_layer.weight|bias``), same prediction semantics

    x' = x + dy_means + dy_std * net(([x, u] - xu_means) / xu_std)

``pred`` / ``pred_batch`` / ``pred_diff`` / ``pred_diff_batch`` call the C ABI
(``ampc_mlp_pred_batch`` / ``ampc_mlp_pred_diff_batch``); the Jacobian is the
analytic chain  W_out prod_l diag(act'(z_l)) W_l  rather than the reference's
autograd over an nx-fold repeated batch (mlp.py:288-295).  Training (mlp.py:177-217,
Adam + SmoothL1 on normalised deltas) stays in PyTorch -- it is outside the MPC
inner loop: sysid/mlp_fit.py runs the reference's loop (its initial weights, its
mini-batch order) on PyTorch-ROCm, for one model or for K models in lockstep, and the
fitted tensors are staged from device memory (ampc_set_mlp_dev).
"""
import numpy as np

from .. import _lib
from .model import Model, ModelFactory

_ACTS = ("relu", "tanh", "sigmoid", "selu")


class MLP(Model):
    def __init__(self, system, n_hidden_layers=3, hidden_size=128, nonlintype="relu",
                 n_train_iters=50, n_batch=64, lr=1e-3, hidden_size_1=None, hidden_size_2=None,
                 hidden_size_3=None, hidden_size_4=None, seed=100, use_cuda=True,
                 precision="f64", device=0):
        super().__init__(system)
        if nonlintype not in _ACTS:
            raise NotImplementedError("Currently supported nonlinearity: relu, selu, tanh, sigmoid")
        n_hidden_layers = int(n_hidden_layers)
        sizes = [int(hidden_size)] * n_hidden_layers
        for i, s in enumerate((hidden_size_1, hidden_size_2, hidden_size_3, hidden_size_4)):
            if s is not None and i < n_hidden_layers:
                sizes[i] = int(s)
        self.hidden_sizes = sizes
        self.nonlintype = nonlintype
        self.seed = seed
        self.precision = precision
        self.device = device
        self._train_data = (n_train_iters, n_batch, lr)
        # jit_kernels: may handles that hold this model start the run-time build of kernels specialised for its
        # shape (3-7 s in the background, cached)?  A tuner sets False on models it fits for ONE evaluation.
        self.jit_kernels = True
        nx, nu = system.obs_dim, system.ctrl_dim
        # torch.nn.Linear default initialisation (U(-1/sqrt(in), 1/sqrt(in))) from a seeded
        # numpy stream; real weights arrive through train() or set_parameters().
        rng = np.random.default_rng(seed)
        dims = [nx + nu] + sizes + [nx]
        self._handle = None
        self._dev_params = None             # (weights, biases, normalisers) as device tensors after a fit
        self._weights, self._biases = [], []
        for fan_in, fan_out in zip(dims[:-1], dims[1:]):
            bound = 1.0 / np.sqrt(fan_in)
            self._weights.append(rng.uniform(-bound, bound, size=(fan_out, fan_in)))
            self._biases.append(rng.uniform(-bound, bound, size=(fan_out,)))
        self.xu_means = np.zeros(nx + nu)
        self.xu_std = np.ones(nx + nu)
        self.dy_means = np.zeros(nx)
        self.dy_std = np.ones(nx)

    # -- reference Model surface (mlp.py:167-175) ---------------------------------
    def traj_to_state(self, traj):
        return traj[-1].obs.copy()

    def update_state(self, state, new_ctrl, new_obs):
        return np.array(new_obs, dtype=np.float64)

    @property
    def state_dim(self):
        return self.system.obs_dim

    # -- device staging -----------------------------------------------------------
    def stage_into(self, handle):
        """Pack the current weights + normalisers into a device handle: from device memory when a fit left
        them on the handle's GPU (ampc_set_mlp_dev: folded and packed by two kernels, no host copy), else
        from the numpy arrays."""
        dp = self._dev_params
        if dp is not None and dp["w"][0].is_cuda and dp["w"][0].device.index == handle.device \
                and self._normalisers_are_the_fit():
            import torch
            torch.cuda.current_stream(dp["w"][0].device).synchronize()     # the fit's kernels have finished
            handle.set_mlp_dev(self.system.obs_dim, self.system.ctrl_dim, self.hidden_sizes,
                               [w.data_ptr() for w in dp["w"]], [b.data_ptr() for b in dp["b"]], self.nonlintype,
                               [v.data_ptr() for v in dp["norm_dev"]])
            return
        handle.set_mlp(self.system.obs_dim, self.system.ctrl_dim, self.weights, self.biases,
                       self.nonlintype, self.xu_means, self.xu_std, self.dy_means, self.dy_std)

    def _normalisers_are_the_fit(self):
        """The device copy stands for the model only while the normalisers are the ones the fit computed."""
        return all(np.array_equal(a, b) for a, b in zip(self._dev_params["norm_np"],
                                                        (self.xu_means, self.xu_std, self.dy_means, self.dy_std)))

    def _dev(self):
        if self._handle is None:
            self._handle = _lib.Handle(self.device, self.precision, jit=getattr(self, "jit_kernels", True))
            self.stage_into(self._handle)
        return self._handle

    def _invalidate(self):
        if self._handle is not None:
            self._handle.close()
        self._handle = None

    def __getstate__(self):
        self._fetch()
        state = self.__dict__.copy()
        state["_handle"] = None          # device handles do not pickle / deepcopy
        state["_dev_params"] = None      # (the copy carries the numpy parameters)
        return state

    # -- inference (HIP) ------------------------------------------------------------
    def pred(self, state, ctrl):
        return self._dev().pred_batch(np.asarray(state)[None, :], np.asarray(ctrl)[None, :])[0]

    def pred_batch(self, states, ctrls):
        return self._dev().pred_batch(states, ctrls)

    def pred_diff(self, state, ctrl):
        o, a, b = self._dev().pred_diff_batch(np.asarray(state)[None, :], np.asarray(ctrl)[None, :])
        return o[0], a[0], b[0]

    def pred_diff_batch(self, states, ctrls):
        return self._dev().pred_diff_batch(states, ctrls)

    # -- parameters (mlp.py:308-321) --------------------------------------------------
    def _state_dict_keys(self):
        keys = []
        for i in range(len(self.hidden_sizes)):
            keys.append(("layers.layer%d.weight" % i, "layers.layer%d.bias" % i))
        keys.append(("output_layer.weight", "output_layer.bias"))
        return keys

    def get_parameters(self):
        net_state = {}
        for (wk, bk), w, b in zip(self._state_dict_keys(), self.weights, self.biases):
            net_state[wk], net_state[bk] = w.copy(), b.copy()
        return {"net_state": net_state, "xu_means": self.xu_means.copy(),
                "xu_std": self.xu_std.copy(), "dy_means": self.dy_means.copy(),
                "dy_std": self.dy_std.copy()}

    @staticmethod
    def _to_numpy(t):
        if hasattr(t, "detach"):
            t = t.detach().cpu().numpy()
        return np.array(t, dtype=np.float64)

    def set_parameters(self, params):
        net = params["net_state"]
        ws, bs = [], []
        for wk, bk in self._state_dict_keys():
            ws.append(self._to_numpy(net[wk]))
            bs.append(self._to_numpy(net[bk]))
        for old, new in zip(self.weights, ws):
            if old.shape != new.shape:
                raise ValueError("net_state layer shape %r does not match model %r"
                                 % (new.shape, old.shape))
        self._weights, self._biases, self._dev_params = ws, bs, None
        self.xu_means = np.array(params["xu_means"], dtype=np.float64)
        self.xu_std = np.array(params["xu_std"], dtype=np.float64)
        self.dy_means = np.array(params["dy_means"], dtype=np.float64)
        self.dy_std = np.array(params["dy_std"], dtype=np.float64)
        self._invalidate()                # weights must be re-staged

    # -- training (PyTorch, outside the hot path; mlp.py:177-217) ------------------------
    def train(self, trajs, silent=False, seed=100):
        """The reference's fit -- normalisers from the data, then n_train_iters epochs of Adam on SmoothL1 over
        shuffled mini-batches, with the reference's initial weights (torch.manual_seed(self.seed) before the
        layers are built) and mini-batch order (torch.manual_seed(seed) + DataLoader(shuffle=True)) -- run by
        sysid/mlp_fit.py on the GPU when there is one (HIP-graph-captured steps), else on the CPU.  The
        fitted parameters stay on the device and are staged from there (ampc_set_mlp_dev)."""
        from .mlp_fit import fit_mlps
        fit_mlps([self], trajs, train_seed=seed)

    def _adopt_fit(self, ws, bs, norms, norms_dev):
        """Take over fitted parameters: float64 torch tensors `ws`, `bs` (any device) plus the data's normalisers
        as numpy arrays and as tensors on the same device."""
        self._invalidate()
        norms = [np.array(v, dtype=np.float64) for v in norms]
        self.xu_means, self.xu_std, self.dy_means, self.dy_std = [v.copy() for v in norms]
        self._dev_params = {"w": list(ws), "b": list(bs), "norm_dev": list(norms_dev), "norm_np": norms}
        self._weights = self._biases = None       # fetched when somebody asks (get_parameters, pickling)

    # weights / biases: numpy [out][in] / [out] per layer.  After a fit they live in device memory and are
    # downloaded on first access; assigning them drops the device copy.
    def _fetch(self):
        if self._weights is None:
            self._weights = [self._to_numpy(w) for w in self._dev_params["w"]]
            self._biases = [self._to_numpy(b) for b in self._dev_params["b"]]

    @property
    def weights(self):
        self._fetch()
        return self._weights

    @weights.setter
    def weights(self, value):
        self._fetch()
        self._weights, self._dev_params = list(value), None
        self._invalidate()

    @property
    def biases(self):
        self._fetch()
        return self._biases

    @biases.setter
    def biases(self, value):
        self._fetch()
        self._biases, self._dev_params = list(value), None
        self._invalidate()


class MLPFactory(ModelFactory):
    """Hyper-parameter space of the reference's MLPFactory (mlp.py:107-135)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.Model = MLP
        self.name = "MLP"

    def get_configuration_space(self):
        try:
            import ConfigSpace as CS
            import ConfigSpace.conditions as CSC
            import ConfigSpace.hyperparameters as CSH
        except ImportError as e:       # ConfigSpace is an optional, tuner-side dependency
            raise ImportError("ConfigSpace is required for get_configuration_space()") from e
        cs = CS.ConfigurationSpace()
        nonlin = CSH.CategoricalHyperparameter("nonlintype", choices=list(_ACTS),
                                               default_value="relu")
        nl = CSH.CategoricalHyperparameter("n_hidden_layers", choices=["1", "2", "3", "4"],
                                           default_value="2")
        hs = [CSH.UniformIntegerHyperparameter("hidden_size_%d" % i, lower=16, upper=256,
                                               default_value=128) for i in (1, 2, 3, 4)]
        lr = CSH.UniformFloatHyperparameter("lr", lower=1e-5, upper=1, default_value=1e-3, log=True)
        cs.add_hyperparameters([nonlin, nl] + hs + [lr])
        cs.add_conditions([
            CSC.InCondition(child=hs[1], parent=nl, values=["2", "3", "4"]),
            CSC.InCondition(child=hs[2], parent=nl, values=["3", "4"]),
            CSC.InCondition(child=hs[3], parent=nl, values=["4"])])
        return cs
