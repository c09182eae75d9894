"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the frozen lip encoder.

Follows the reference's `FRCNNVideoModel.forward` with the ResNet-18 trunk in eval mode:
  frontend3D  src/models/videomodels/frcnn_videomodel.py:43-55  Conv3d(1->64,(5,7,7),s(1,2,2),p(2,3,3),no bias) -> BatchNorm3d ->
              PReLU(64) (or ReLU) -> MaxPool3d((1,3,3),(1,2,2),(0,1,1))
  3D -> 2D    frcnn_videomodel.py:10-13, 62-64                  [B,C,T,H,W] -> [B*T,C,H,W]
  BasicBlock  resnet.py:49-66                                   conv3x3(stride)-bn-act-conv3x3-bn, (+ 1x1(stride)-bn shortcut), add, act
  ResNet      resnet.py:119-126                                 layer1..4, AdaptiveAvgPool2d(1), flatten
  output      frcnn_videomodel.py:66                            view(B,T,512).transpose(1,2)
Works on a reference-keyed state dict; plain torch CPU arithmetic in the dtype of the weights (float64 for the checker).
Pinned by tests/golden/lip.npz, produced by RUNNING the reference (oracle/gen_golden_lip.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EPS = 1e-5  # nn.BatchNorm default


def _bn(x, sd, pre):
    shape = [1, -1] + [1] * (x.dim() - 2)
    scale = sd[pre + ".weight"] / torch.sqrt(sd[pre + ".running_var"] + EPS)
    return (x - sd[pre + ".running_mean"].view(shape)) * scale.view(shape) + sd[pre + ".bias"].view(shape)


def _act(x, sd, pre):
    key = pre + ".weight"
    if key in sd:  # PReLU(planes)
        a = sd[key].view([1, -1] + [1] * (x.dim() - 2))
        return torch.where(x >= 0, x, a * x)
    return x.clamp_min(0)  # ReLU


def _block(x, sd, pre, stride):
    out = F.conv2d(x, sd[pre + ".conv1.weight"], stride=stride, padding=1)
    out = _act(_bn(out, sd, pre + ".bn1"), sd, pre + ".relu1")
    out = _bn(F.conv2d(out, sd[pre + ".conv2.weight"], padding=1), sd, pre + ".bn2")
    if pre + ".downsample.0.weight" in sd:
        x = _bn(F.conv2d(x, sd[pre + ".downsample.0.weight"], stride=stride), sd, pre + ".downsample.1")
    return _act(out + x, sd, pre + ".relu2")


def frcnn_forward(sd: dict, x: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    """x [B,1,T,H,W] -> [B,512,T]; `taps` (optional dict) receives the stage boundaries in NCHW."""
    dt = sd["frontend3D.0.weight"].dtype
    x = x.to(dt)
    B, _, T = x.shape[:3]
    y = F.conv3d(x, sd["frontend3D.0.weight"], stride=(1, 2, 2), padding=(2, 3, 3))
    y = _act(_bn(y, sd, "frontend3D.1"), sd, "frontend3D.2")
    y = F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    y = y.transpose(1, 2).reshape(B * T, y.shape[1], y.shape[3], y.shape[4])
    if taps is not None:
        taps["front"] = y
    for li in range(1, 5):
        for bi in range(2):
            y = _block(y, sd, f"trunk.layer{li}.{bi}", 2 if (li > 1 and bi == 0) else 1)
        if taps is not None:
            taps[f"layer{li}"] = y
    y = y.mean(dim=(2, 3))
    return y.view(B, T, -1).transpose(1, 2).contiguous()


from rtfs_net_amd.synthetic import lip_inputs  # noqa: E402,F401  (generator shared with bench.py; kept importable from here for the fixture scripts)
