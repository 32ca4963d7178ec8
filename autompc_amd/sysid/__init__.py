from .model import Model, ModelFactory
from .mlp import MLP, MLPFactory
from .sindy import SINDy, SINDyFactory
from .linear import ARX, ARXFactory, Koopman, KoopmanFactory

__all__ = ["Model", "ModelFactory", "MLP", "MLPFactory", "SINDy", "SINDyFactory", "ARX",
           "ARXFactory", "Koopman", "KoopmanFactory"]
