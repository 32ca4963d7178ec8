"""System-ID model plugin interface.

What the controllers, ``simulate`` and the tuner call on a dynamics model (reference:
autompc/sysid/model.py:8-53 factory, :55-244 model):

    state  = model.traj_to_state(traj)                  model state from history
    state  = model.update_state(state, ctrl, new_obs)   fold in a new observation
    x'     = model.pred(state, ctrl)                    one step
    X'     = model.pred_batch(states, ctrls)            N steps at once
    x',A,B = model.pred_diff(state, ctrl)               step + Jacobians
    X',A,B = model.pred_diff_batch(states, ctrls)
    model.state_dim / is_diff / is_linear / to_linear() / get_parameters() / set_parameters()

``pred_parallel`` / ``pred_diff_parallel`` are the newer-upstream names of the batched calls
 and are provided as aliases.  The batched defaults below loop over the single-row
calls, as the reference's do; the device models override them with one launch.
"""
import abc

import numpy as np


class ModelFactory(abc.ABC):
    """Builds (and trains) models from a configuration.  Subclasses provide ``Model`` (the class)
    and ``name``; keyword arguments given to the factory override configuration entries."""

    def __init__(self, system, **kwargs):
        self.system, self.kwargs = system, kwargs

    def __call__(self, cfg, train_trajs, silent=False, skip_train_model=False):
        settings = {**dict(cfg.get_dictionary()), **self.kwargs}
        model = self.Model(self.system, **settings)
        model.factory = self
        if not skip_train_model:
            model.train(train_trajs, silent=silent)
        return model

    @abc.abstractmethod
    def get_configuration_space(self):
        """ConfigSpace with the model's hyper-parameters (optional dependency)."""


class Model(abc.ABC):
    def __init__(self, system):
        self.system = system

    # -- state ------------------------------------------------------------------------------
    @property
    @abc.abstractmethod
    def state_dim(self):
        """Length of the model state vector (>= system.obs_dim)."""

    @abc.abstractmethod
    def traj_to_state(self, traj):
        """Model state after observing `traj`."""

    @abc.abstractmethod
    def update_state(self, state, new_ctrl, new_obs):
        """Model state after applying `new_ctrl` in `state` and then observing `new_obs`."""

    # -- prediction -------------------------------------------------------------------------
    @abc.abstractmethod
    def pred(self, state, ctrl):
        """Next model state."""

    def pred_batch(self, states, ctrls):
        rows = [self.pred(s, c) for s, c in zip(states, ctrls)]
        return np.array(rows).reshape(len(rows), self.state_dim)

    def pred_diff(self, state, ctrl):
        """(next state, d next / d state, d next / d ctrl); differentiable models override."""
        raise NotImplementedError

    def pred_diff_batch(self, states, ctrls):
        n, ns, nu = len(states), self.state_dim, self.system.ctrl_dim
        nxt, jac_x, jac_u = np.empty((n, ns)), np.empty((n, ns, ns)), np.empty((n, ns, nu))
        for k, (s, c) in enumerate(zip(states, ctrls)):
            nxt[k], jac_x[k], jac_u[k] = self.pred_diff(s, c)
        return nxt, jac_x, jac_u

    def pred_parallel(self, states, ctrls):
        return self.pred_batch(states, ctrls)

    def pred_diff_parallel(self, states, ctrls):
        return self.pred_diff_batch(states, ctrls)

    def to_linear(self):
        """(A, B) of x' = A x + B u; linear models override."""
        raise NotImplementedError

    # -- capabilities: derived from what the subclass overrides -------------------------------
    @property
    def is_diff(self):
        return type(self).pred_diff is not Model.pred_diff

    @property
    def is_linear(self):
        return type(self).to_linear is not Model.to_linear

    # -- fitting / persistence ----------------------------------------------------------------
    def train(self, trajs, silent=False):
        raise NotImplementedError

    def get_parameters(self):
        raise NotImplementedError

    def set_parameters(self, params):
        raise NotImplementedError
