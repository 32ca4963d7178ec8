"""Controller plugin interface (reference: autompc/control/controller.py:6-121).

``run(state, new_obs) -> (ctrl, newstate)``, ``traj_to_state(traj)``, ``reset()``
and the ``state_dim`` property are the whole surface ``simulate`` and the tuner
use.  ``step`` is the newer-upstream spelling of ``run``.
"""
from abc import ABC, abstractmethod


class ControllerFactory(ABC):
    def __init__(self, system, **kwargs):
        self.system = system
        self.kwargs = kwargs

    def __call__(self, cfg, task, model):
        controller_kwargs = dict(cfg.get_dictionary())
        controller_kwargs.update(self.kwargs)
        return self.Controller(self.system, task, model, **controller_kwargs)

    def get_configuration_space(self):
        raise NotImplementedError


class Controller(ABC):
    def __init__(self, system, task, model):
        self.system = system
        self.model = model
        self.task = task

    @abstractmethod
    def traj_to_state(self, traj):
        raise NotImplementedError

    @abstractmethod
    def run(self, state, new_obs):
        raise NotImplementedError

    def step(self, state, new_obs):
        return self.run(state, new_obs)

    def reset(self):
        pass

    @property
    @abstractmethod
    def state_dim(self):
        raise NotImplementedError
