"""Mouth-ROI preprocessing on the GPU.

The reference prepares the lip stream on the host, per clip, in numpy (src/datas/transform.py:151-167, called from
src/datas/avspeech_dataset.py:44,124): uint8 grey ROI frames `[T, H, W]` -> `Normalize(0, 255)` -> `CenterCrop(88, 88)` (val / test)
or `RandomCrop(88, 88)` + `HorizontalFlip(0.5)` (train) -> `Normalize(0.421, 0.165)` -> float.  Here the uint8 frames go to the GPU
as they are (a quarter of the bytes) and ONE kernel (`rtfs_lip_roi_fwd`) produces the zero-padded float clip the lip encoder's stem
reads.  The per-value map is a 256-entry table evaluated on the host in float64 exactly as numpy evaluates the pipeline, so the
result is bit-identical to the reference's.  Random crops / flips are drawn on the host (one triple per clip) and passed in.
"""
from __future__ import annotations

import random

import numpy as np
import torch

try:
    from ... import lib
except ImportError:  # relocated copy of the models sub-package (train.py:95 / test.py:33-36): binding from the installed package
    from rtfs_net_amd import lib


class MouthROI:
    def __init__(self, crop_size=(88, 88), mean=0.421, std=0.165):
        self.crop_size = (int(crop_size[0]), int(crop_size[1]))
        self.mean, self.std = mean, std
        u = np.arange(256, dtype=np.uint8)
        self._lut_host = torch.from_numpy(((((u - 0.0) / 255.0) - mean) / std).astype(np.float32))  # Normalize(0,255) then Normalize(mean,std)
        self._lut = {}

    def _table(self, dev):
        if dev not in self._lut:
            self._lut[dev] = self._lut_host.to(dev)
        return self._lut[dev]

    def center_offsets(self, H: int, W: int):
        """CenterCrop's offsets (transform.py:96-101)."""
        th, tw = self.crop_size
        return int(round((H - th)) / 2.0), int(round((W - tw)) / 2.0)

    def random_crops(self, B: int, H: int, W: int, flip_ratio=0.5, rng: random.Random = None) -> torch.Tensor:
        """One (dy, dx, flip) per clip, drawn like RandomCrop + HorizontalFlip (transform.py:114-147): w offset first, then h, then the
        flip coin, from Python's `random` (or the given `random.Random`)."""
        rng = rng or random
        th, tw = self.crop_size
        rows = []
        for _ in range(B):
            dx = rng.randint(0, W - tw)
            dy = rng.randint(0, H - th)
            rows.append((dy, dx, 1 if rng.random() < flip_ratio else 0))
        return torch.tensor(rows, dtype=torch.int32)

    def padded(self, rois: torch.Tensor, crops: torch.Tensor = None) -> torch.Tensor:
        """rois uint8 [B, T, H, W] on the GPU -> zero-padded normalised clip [B, T+4, ch+6, cw+6] float32."""
        if rois.dtype != torch.uint8 or rois.dim() != 4:
            raise ValueError(f"expected uint8 ROIs [B, T, H, W], got {rois.dtype} {tuple(rois.shape)}")
        if not rois.is_cuda:
            raise RuntimeError("MouthROI runs on the HIP path only (no CPU fallback)")
        B, T, H, W = rois.shape
        ch, cw = self.crop_size
        if H < ch or W < cw:
            raise ValueError(f"ROI {H}x{W} smaller than the crop {ch}x{cw}")
        if crops is not None:
            crops = crops.to(device=rois.device, dtype=torch.int32).contiguous()
            if crops.shape != (B, 3):
                raise ValueError("crops must be [B, 3] = (dy, dx, flip)")
            c = crops.cpu()
            if int(c[:, 0].min()) < 0 or int(c[:, 0].max()) > H - ch or int(c[:, 1].min()) < 0 or int(c[:, 1].max()) > W - cw:
                raise ValueError("crop offsets outside the ROI")
        P = torch.empty(B, T + 4, ch + 6, cw + 6, device=rois.device)
        lib.call("rtfs_lip_roi_fwd", rois.contiguous(), crops, self._table(rois.device), P, B, T, H, W, ch, cw)
        return P

    def __call__(self, rois: torch.Tensor, crops: torch.Tensor = None) -> torch.Tensor:
        """-> [B, 1, T, ch, cw] float32: what the reference's DataLoader hands to the video model (avspeech_dataset.py:137, core.py:89)."""
        return self.padded(rois, crops)[:, 2:-2, 3:-3, 3:-3].unsqueeze(1).contiguous()
