from .adain_model import AdaINModel
from .base_model import BaseModel
from .model import Model

__all__ = ["AdaINModel", "BaseModel", "Model"]
