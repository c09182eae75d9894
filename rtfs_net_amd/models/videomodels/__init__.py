"""Host-side mirror of the reference's `src.models.videomodels` package (src/models/videomodels/__init__.py:8-50):
`FRCNNVideoModel` (the frozen lip encoder), `ResNet`, `BasicBlock`, `update_frcnn_parameter`,
case-insensitive `get`, `register_model`.  Only the ResNet-18 backbone of the shipped configs is built."""
from .frcnn_videomodel import BasicBlock, FRCNNVideoModel, ResNet, update_frcnn_parameter
from .roi import MouthROI

__all__ = ["ResNet", "BasicBlock", "FRCNNVideoModel", "update_frcnn_parameter", "MouthROI", "get", "register_model"]


def register_model(custom_model):
    """Register a custom model, gettable with `videomodels.get` (src/models/videomodels/__init__.py:22-31)."""
    if custom_model.__name__ in globals().keys() or custom_model.__name__.lower() in globals().keys():
        raise ValueError(f"Model {custom_model.__name__} already exists. Choose another name.")
    globals().update({custom_model.__name__: custom_model})


def get(identifier):
    """Model class from a (case-insensitive) name (src/models/videomodels/__init__.py:34-50)."""
    if isinstance(identifier, str):
        cls = {k.lower(): v for k, v in globals().items()}.get(identifier.lower())
        if cls is None:
            raise ValueError(f"Could not interpret model name : {str(identifier)}")
        return cls
    raise ValueError(f"Could not interpret model name : {str(identifier)}")
