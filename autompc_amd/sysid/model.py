"""System-ID model plugin interface.

Same contract as the reference's ``Model`` / ``ModelFactory`` ABCs (reference:
autompc/sysid/model.py:8-53 factory, :55-244 model): controllers call
``traj_to_state``, ``update_state``, ``pred``, ``pred_batch``, ``pred_diff``,
``pred_diff_batch`` and read ``state_dim`` / ``is_diff`` / ``is_linear``.
``pred_parallel`` / ``pred_diff_parallel`` are the newer-upstream names for the
batched calls and are provided as aliases.
"""
from abc import ABC, abstractmethod

import numpy as np


class ModelFactory(ABC):
    def __init__(self, system, **kwargs):
        self.system = system
        self.kwargs = kwargs

    def __call__(self, cfg, train_trajs, silent=False, skip_train_model=False):
        model_args = dict(cfg.get_dictionary())
        model_args.update(self.kwargs)
        model = self.Model(self.system, **model_args)
        model.factory = self
        if not skip_train_model:
            model.train(train_trajs, silent=silent)
        return model

    @abstractmethod
    def get_configuration_space(self):
        raise NotImplementedError


class Model(ABC):
    def __init__(self, system):
        self.system = system

    @abstractmethod
    def traj_to_state(self, traj):
        raise NotImplementedError

    @abstractmethod
    def update_state(self, state, new_ctrl, new_obs):
        raise NotImplementedError

    @abstractmethod
    def pred(self, state, ctrl):
        raise NotImplementedError

    def pred_batch(self, states, ctrls):
        out = np.empty((states.shape[0], self.state_dim))
        for i in range(states.shape[0]):
            out[i, :] = self.pred(states[i, :], ctrls[i, :])
        return out

    def pred_diff(self, state, ctrl):
        raise NotImplementedError

    def pred_diff_batch(self, states, ctrls):
        m, n = states.shape[0], self.state_dim
        out = np.empty((m, n))
        jx = np.empty((m, n, n))
        ju = np.empty((m, n, self.system.ctrl_dim))
        for i in range(m):
            out[i], jx[i], ju[i] = self.pred_diff(states[i, :], ctrls[i, :])
        return out, jx, ju

    # newer-upstream spellings
    def pred_parallel(self, states, ctrls):
        return self.pred_batch(states, ctrls)

    def pred_diff_parallel(self, states, ctrls):
        return self.pred_diff_batch(states, ctrls)

    def to_linear(self):
        raise NotImplementedError

    def train(self, trajs, silent=False):
        raise NotImplementedError

    def get_parameters(self):
        raise NotImplementedError

    def set_parameters(self, params):
        raise NotImplementedError

    @property
    @abstractmethod
    def state_dim(self):
        raise NotImplementedError

    @property
    def is_linear(self):
        return type(self).to_linear is not Model.to_linear

    @property
    def is_diff(self):
        return type(self).pred_diff is not Model.pred_diff
