"""Host-side mirror of the reference's `src.models.videomodels` package: `FRCNNVideoModel` (the frozen lip encoder),
`ResNet`, `BasicBlock`, `update_frcnn_parameter`, `MouthROI` (GPU mouth-ROI preprocessing, §8 f4), case-insensitive `get`,
`register_model`.  Only the ResNet-18 backbone of the shipped configs is built."""
from .._registry import make_registry
from .frcnn_videomodel import BasicBlock, FRCNNVideoModel, ResNet, update_frcnn_parameter
from .roi import MouthROI

__all__ = ["ResNet", "BasicBlock", "FRCNNVideoModel", "update_frcnn_parameter", "MouthROI", "get", "register_model"]

register_model, get = make_registry(globals())
