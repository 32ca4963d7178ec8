from .controller import Controller, ControllerFactory
from .mppi import MPPI, MPPIFactory

__all__ = ["Controller", "ControllerFactory", "MPPI", "MPPIFactory"]
