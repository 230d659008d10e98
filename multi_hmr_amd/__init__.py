"""MI355X-native Multi-HMR batched inference (drop-in for naver/multi-hmr ``model.Model`` / ``demo.forward_model``)."""
from .model import Model  # noqa: F401
from .demo import forward_model, get_camera_parameters, load_model, open_image  # noqa: F401

from .preprocess import Preprocessor  # noqa: F401
from .graphed import GraphedForward  # noqa: F401

__all__ = ["Model", "GraphedForward", "Preprocessor", "forward_model", "get_camera_parameters", "load_model", "open_image"]
