"""Controller plugin interface.

The surface ``simulate``, ``Pipeline`` and the tuner use (reference:
autompc/control/controller.py:6-121) and nothing more:

    controller = factory(cfg, task, model)        # ControllerFactory.__call__
    state = controller.traj_to_state(traj)        # controller state from history
    ctrl, state = controller.run(state, new_obs)  # one control step
    controller.reset()                            # back to construction-time state
    controller.state_dim                          # length of `state`

``step`` is the newer-upstream spelling of ``run`` and is provided as an alias.
Controllers are stateful and single-threaded, like the reference's.
"""
import abc


class ControllerFactory(abc.ABC):
    """Builds controllers from a configuration.  Subclasses set ``Controller`` (the class) and
    ``name``; keyword arguments given to the factory override configuration entries."""

    def __init__(self, system, **kwargs):
        self.system, self.kwargs = system, kwargs

    def __call__(self, cfg, task, model):
        settings = {**dict(cfg.get_dictionary()), **self.kwargs}
        return self.Controller(self.system, task, model, **settings)

    def get_configuration_space(self):
        """ConfigSpace with the controller's hyper-parameters (optional dependency)."""
        raise NotImplementedError


class Controller(abc.ABC):
    def __init__(self, system, task, model):
        self.system, self.task, self.model = system, task, model

    @property
    @abc.abstractmethod
    def state_dim(self):
        """Length of the controller state vector."""

    @abc.abstractmethod
    def traj_to_state(self, traj):
        """Controller state after observing `traj` (numpy vector of length state_dim)."""

    @abc.abstractmethod
    def run(self, state, new_obs):
        """One control step: (control to apply, updated controller state)."""

    def step(self, state, new_obs):
        return self.run(state, new_obs)

    def reset(self):
        """Restore construction-time state (default: stateless)."""
