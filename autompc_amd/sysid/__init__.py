from .model import Model, ModelFactory
from .mlp import MLP, MLPFactory
from .sindy import SINDy, SINDyFactory

__all__ = ["Model", "ModelFactory", "MLP", "MLPFactory", "SINDy", "SINDyFactory"]
