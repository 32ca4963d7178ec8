from .controller import Controller, ControllerFactory
from .mppi import MPPI, MPPIFactory
from .ilqr import IterativeLQR, IterativeLQRFactory

__all__ = ["Controller", "ControllerFactory", "MPPI", "MPPIFactory", "IterativeLQR",
           "IterativeLQRFactory"]
