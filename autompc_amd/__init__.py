"""autompc_amd: MI355X-native MPC inner-solve path behind AutoMPC's plugin surface.

Host-side types (System / Task / Trajectory / costs) are importable anywhere;
everything that computes (MLP model, MPPI, iLQR, batch evaluator) goes through
the C-ABI library ``libautompc_hip.so`` and raises if it is not built.
"""
from .system import System
from .trajectory import Trajectory, TimeStep, zeros, empty, extend
from .task import Task
from .costs import Cost, QuadCost, SumCost, ThresholdCost, BoxThresholdCost

from .sysid import (Model, ModelFactory, MLP, MLPFactory, SINDy, SINDyFactory, ARX, ARXFactory,
                    Koopman, KoopmanFactory)
from .control import (Controller, ControllerFactory, MPPI, MPPIFactory, IterativeLQR,
                      IterativeLQRFactory)
from .utils import simulate

__all__ = ["Model", "ModelFactory", "MLP", "MLPFactory", "SINDy", "SINDyFactory", "ARX", "ARXFactory", "Koopman", "KoopmanFactory", "Controller", "ControllerFactory",
           "MPPI", "MPPIFactory", "IterativeLQR", "IterativeLQRFactory", "simulate", "System", "Trajectory", "TimeStep", "zeros", "empty", "extend", "Task",
           "Cost", "QuadCost", "SumCost", "ThresholdCost", "BoxThresholdCost"]
