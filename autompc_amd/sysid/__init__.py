from .model import Model, ModelFactory
from .mlp import MLP, MLPFactory

__all__ = ["Model", "ModelFactory", "MLP", "MLPFactory"]
