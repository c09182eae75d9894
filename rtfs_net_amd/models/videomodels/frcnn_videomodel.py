"""Frozen lip encoder on the HIP path.

Mirrors `FRCNNVideoModel` of the reference (src/models/videomodels/frcnn_videomodel.py:16-72): constructor arguments, sub-module
names (`frontend3D.{0,1,2}`, `trunk.layer{1..4}.{0,1}.{conv1,bn1,relu1,conv2,bn2,relu2,downsample.{0,1}}`) and therefore the
state-dict keys of the pretrained lip-reading checkpoints, `init_from`, BatchNorm frozen in `train()`, `[B, 1, T, H, W]` mouth
crops in, `[B, 512, T]` embeddings out.  The torch modules below only OWN the parameters; the arithmetic runs in
`csrc/lip.hip` (implicit-GEMM convolutions on the fp32 MFMA pipe, BatchNorm folded on the host, PReLU / residual in the
epilogue).  There is no PyTorch fallback: without the HIP library or off the GPU the forward raises.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

try:
    from ... import lib
except ImportError:  # relocated copy of the models sub-package (train.py:95 / test.py:33-36): binding from the installed package
    from rtfs_net_amd import lib


def _act(relu_type: str, planes: int) -> nn.Module:
    if relu_type == "relu":
        return nn.ReLU(inplace=True)
    if relu_type == "prelu":
        return nn.PReLU(num_parameters=planes)
    raise Exception("relu type not implemented")


class BasicBlock(nn.Module):
    """Parameter holder of one residual block (resnet.py:27-66): conv3x3-bn-act-conv3x3-bn (+ downsample) + add + act."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="relu"):
        super().__init__()
        assert relu_type in ["relu", "prelu"]
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1 = _act(relu_type, planes)
        self.relu2 = _act(relu_type, planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("BasicBlock holds parameters only; run it through FRCNNVideoModel (HIP path)")


class ResNet(nn.Module):
    """Parameter holder of the trunk (resnet.py:68-130): four stages of `layers[i]` blocks, 64/128/256/512 planes, stride 2 from
    stage 2 on with a 1x1-conv + BatchNorm shortcut, global average pool."""

    def __init__(self, block, layers, num_classes=1000, relu_type="relu", gamma_zero=False, avg_pool_downsample=False):
        super().__init__()
        if avg_pool_downsample:
            raise ValueError("avg_pool_downsample=True is not built (the lip encoder of the shipped configs does not use it)")
        self.inplanes = 64
        self.relu_type = relu_type
        self.gamma_zero = gamma_zero
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():  # the reference's default init (resnet.py:91-104)
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if gamma_zero:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    m.bn2.weight.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        seq = [block(self.inplanes, planes, stride, downsample, relu_type=self.relu_type)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, relu_type=self.relu_type) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        raise RuntimeError("ResNet holds parameters only; run it through FRCNNVideoModel (HIP path)")


def _fold(conv_w: torch.Tensor, bn: nn.modules.batchnorm._BatchNorm):
    """BatchNorm (eval) folded into the convolution: rows scaled by gamma / sqrt(var + eps), bias = beta - mean * scale."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    w = conv_w.detach().float() * scale.view(-1, *([1] * (conv_w.ndim - 1)))
    return w, shift.contiguous()


def _slopes(act: nn.Module, planes: int, dev) -> torch.Tensor:
    if isinstance(act, nn.PReLU):
        return act.weight.detach().float().expand(planes).contiguous()
    return torch.zeros(planes, device=dev)  # ReLU = PReLU with slope 0


class _LipWeights:
    """Device-side operands of csrc/lip.hip, rebuilt when a parameter or running statistic changes."""

    def __init__(self, model: "FRCNNVideoModel"):
        conv3d, bn3d, act = model.frontend3D[0], model.frontend3D[1], model.frontend3D[2]
        dev = conv3d.weight.device
        w, self.stem_bias = _fold(conv3d.weight, bn3d)  # [64, 1, 5, 7, 7]
        ws = torch.zeros(64, 256, device=dev)
        ws[:, :245] = w.reshape(64, 245)
        self.stem_w = ws.contiguous()
        self.stem_slope = _slopes(act, 64, dev)
        self.blocks = []
        for layer in (model.trunk.layer1, model.trunk.layer2, model.trunk.layer3, model.trunk.layer4):
            for blk in layer:
                planes = blk.conv1.out_channels
                w1, b1 = _fold(blk.conv1.weight, blk.bn1)
                w2, b2 = _fold(blk.conv2.weight, blk.bn2)
                ent = {
                    "stride": blk.stride, "cin": blk.conv1.in_channels, "cout": planes,
                    "w1": w1.permute(0, 2, 3, 1).reshape(planes, -1).contiguous(), "b1": b1, "s1": _slopes(blk.relu1, planes, dev),
                    "w2": w2.permute(0, 2, 3, 1).reshape(planes, -1).contiguous(), "b2": b2, "s2": _slopes(blk.relu2, planes, dev),
                    "wd": None, "bd": None,
                }
                if blk.downsample is not None:
                    wd, bd = _fold(blk.downsample[0].weight, blk.downsample[1])
                    ent["wd"], ent["bd"] = wd.permute(0, 2, 3, 1).reshape(planes, -1).contiguous(), bd
                self.blocks.append(ent)
        self.tapoff = {}

    def taps(self, H: int, W: int, dev) -> torch.Tensor:
        if (H, W) not in self.tapoff:
            k = torch.arange(245)
            dt, dy, dx = k // 49, (k // 7) % 7, k % 7
            off = torch.zeros(256, dtype=torch.int32)
            off[:245] = ((dt * (H + 6) + dy) * (W + 6) + dx).to(torch.int32)
            self.tapoff[(H, W)] = off.to(dev)
        return self.tapoff[(H, W)]


class FRCNNVideoModel(nn.Module):
    def __init__(self, backbone_type="resnet", relu_type="prelu", width_mult=1.0, pretrain=None, print_macs=True, *args, **kwargs):
        super().__init__()
        if backbone_type != "resnet":
            raise ValueError(f"backbone_type={backbone_type!r} is not built: the HIP lip encoder covers the ResNet-18 trunk "
                             "(src/models/videomodels/frcnn_videomodel.py:30-33)")
        self.backbone_type = backbone_type
        self.frontend_nout = 64
        self.backend_out = 512
        self.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type=relu_type)
        frontend_relu = nn.PReLU(num_parameters=self.frontend_nout) if relu_type == "prelu" else nn.ReLU()
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, self.frontend_nout, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(self.frontend_nout),
            frontend_relu,
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)),
        )
        self.pretrain = pretrain
        self._lw = None
        self._lw_key = None
        if pretrain:
            self.init_from(pretrain)
        if print_macs:
            self.get_MACs()

    # ---- the HIP forward -------------------------------------------------------------------------
    def _weights(self) -> _LipWeights:
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        if self._lw is None or key != self._lw_key:
            self._lw, self._lw_key = _LipWeights(self), key
        return self._lw

    def forward(self, x: torch.Tensor):
        if x.dim() != 5 or x.shape[1] != 1:
            raise ValueError(f"expected mouth crops [B, 1, T, H, W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("FRCNNVideoModel runs on the HIP path only (no CPU fallback): move the model and the input to the GPU")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("the HIP lip encoder is inference-only (frozen encoder, core.py:87-89): call it under torch.no_grad() "
                               "or freeze its parameters")
        B, _, T, H, W = x.shape
        P = F.pad(x[:, 0].float(), (3, 3, 3, 3, 2, 2)).contiguous()  # the conv's zero padding, materialised once (H+6, W+6, T+4)
        emb = self._encode_padded(P, B, T, H, W)
        return emb.to(x.dtype) if x.dtype != torch.float32 else emb

    def forward_rois(self, rois: torch.Tensor, crops: torch.Tensor = None, crop_size=(88, 88), mean=0.421, std=0.165):
        """uint8 mouth ROIs [B, T, H, W] (the `.npz["data"]` arrays of the dataset, avspeech_dataset.py:121-124) -> [B, 512, T]:
        the reference's preprocessing pipeline (transform.py:151-167) runs fused into the stem's padded input, see `MouthROI`."""
        from .roi import MouthROI

        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("the HIP lip encoder is inference-only: call it under torch.no_grad() or freeze its parameters")
        P = MouthROI(crop_size, mean, std).padded(rois, crops)
        return self._encode_padded(P, rois.shape[0], rois.shape[1], crop_size[0], crop_size[1])

    def _encode_padded(self, P: torch.Tensor, B: int, T: int, H: int, W: int) -> torch.Tensor:
        """P: zero-padded clip [B, T+4, H+6, W+6] float32 on the GPU."""
        w = self._weights()
        N = B * T
        dev = P.device
        Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        c1 = torch.empty(N, Hc, Wc, 64, device=dev)
        lib.call("rtfs_lip_stem_fwd", P, w.stem_w, w.taps(H, W, dev), w.stem_bias, w.stem_slope, c1, B, T, H, W)
        h, wd_ = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
        cur = torch.empty(N, h, wd_, 64, device=dev)
        lib.call("rtfs_lip_maxpool_fwd", c1, cur, N, Hc, Wc)
        del c1
        for b in w.blocks:
            s, cin, cout = b["stride"], b["cin"], b["cout"]
            ho, wo = (h - 1) // s + 1, (wd_ - 1) // s + 1
            y1 = torch.empty(N, ho, wo, cout, device=dev)
            lib.call("rtfs_conv_nhwc_fwd", cur, b["w1"], b["b1"], b["s1"], None, y1, N, h, wd_, cin, cout, 3, s)
            if b["wd"] is not None:
                res = torch.empty(N, ho, wo, cout, device=dev)
                lib.call("rtfs_conv_nhwc_fwd", cur, b["wd"], b["bd"], None, None, res, N, h, wd_, cin, cout, 1, s)
            else:
                res = cur
            out = torch.empty(N, ho, wo, cout, device=dev)
            lib.call("rtfs_conv_nhwc_fwd", y1, b["w2"], b["b2"], b["s2"], res, out, N, ho, wo, cout, cout, 3, 1)
            cur, h, wd_ = out, ho, wo
        emb = torch.empty(B, self.backend_out, T, device=dev)
        lib.call("rtfs_lip_avgpool_fwd", cur, emb, B, T, h * wd_, self.backend_out)
        return emb

    # ---- reference protocol ----------------------------------------------------------------------
    def init_from(self, path):
        pretrained_dict = torch.load(path, map_location="cpu")["model_state_dict"]
        update_frcnn_parameter(self, pretrained_dict)

    def train(self, mode=True):
        super().train(mode)
        if mode:  # BatchNorm statistics stay frozen (frcnn_videomodel.py:75-80)
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def get_MACs(self):
        """MACs of one 2-s clip of 88x88 crops, counted analytically per convolution (the reference profiles with thop,
        frcnn_videomodel.py:82-98); sets `.macs` (millions) and `.number_of_parameters` (thousands) and prints them."""
        T, h, w = 50, 88, 88
        hc, wc = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        macs = T * hc * wc * 64 * 245
        h, w = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
        for layer in (self.trunk.layer1, self.trunk.layer2, self.trunk.layer3, self.trunk.layer4):
            for blk in layer:
                cin, cout, s = blk.conv1.in_channels, blk.conv1.out_channels, blk.stride
                h, w = (h - 1) // s + 1, (w - 1) // s + 1
                macs += T * h * w * cout * (9 * cin + 9 * cout + (cin if blk.downsample is not None else 0))
        self.macs = macs / 1000000
        self.number_of_parameters = sum(p.numel() for p in self.parameters()) / 1000
        print("Pretrained Video Backbone\nNumber of MACs: {:,.1f}M\nNumber of parameters: {:,.1f}K\n".format(self.macs, self.number_of_parameters))


def update_frcnn_parameter(model, pretrained_dict):
    """Load a lip-reading checkpoint minus its temporal head (keys containing "tcn") and freeze everything
    (frcnn_videomodel.py:101-115)."""
    model_dict = model.state_dict()
    model_dict.update({k: v for k, v in pretrained_dict.items() if "tcn" not in k})
    model.load_state_dict(model_dict)
    for p in model.parameters():
        p.requires_grad = False
    return model
