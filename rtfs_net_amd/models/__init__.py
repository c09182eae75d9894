"""Host-side mirror of the reference's `src.models` package: `AVNet` (alias `RTFSNet`), the `videomodels` sub-package (train.py:17),
case-insensitive `get`, `register_model` (behaviour: see `_registry.py`)."""
from . import videomodels
from ._registry import make_registry
from .avnet import AVNet

RTFSNet = AVNet

__all__ = ["AVNet", "RTFSNet", "videomodels", "get", "register_model"]

register_model, get = make_registry(globals())
