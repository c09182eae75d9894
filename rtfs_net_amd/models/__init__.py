"""Host-side mirror of the reference's `src.models` package (src/models/__init__.py:8-42):
`AVNet` (alias `RTFSNet`), the `videomodels` sub-package (train.py:17), case-insensitive `get`, `register_model`."""
from . import videomodels
from .avnet import AVNet

RTFSNet = AVNet

__all__ = ["AVNet", "RTFSNet", "videomodels", "get", "register_model"]


def register_model(custom_model):
    """Register a custom model, gettable with `models.get` (src/models/__init__.py:15-25)."""
    if custom_model.__name__ in globals().keys() or custom_model.__name__.lower() in globals().keys():
        raise ValueError(f"Model {custom_model.__name__} already exists. Choose another name.")
    globals().update({custom_model.__name__: custom_model})


def get(identifier):
    """Model class from a (case-insensitive) name (src/models/__init__.py:28-42)."""
    if isinstance(identifier, str):
        cls = {k.lower(): v for k, v in globals().items()}.get(identifier.lower())
        if cls is None:
            raise ValueError(f"Could not interpret model name : {str(identifier)}")
        return cls
    raise ValueError(f"Could not interpret model name : {str(identifier)}")
