"""ORACLE (test infrastructure): numpy restatement of the reference's mouth-ROI preprocessing pipelines
(src/datas/transform.py:151-167): Normalize(0, 255) -> CenterCrop | RandomCrop + HorizontalFlip -> Normalize(0.421, 0.165), evaluated in
float64 as numpy does for a uint8 input, then cast to float32 where the reference casts (`mouth.type_as(wav)`, src/system/core.py:89).
Pinned by tests/golden/roi.npz (oracle/gen_golden_roi.py runs the reference's own classes)."""
from __future__ import annotations

import random

import numpy as np

MEAN, STD = 0.421, 0.165


def preprocess(frames: np.ndarray, mode: str = "val", crop=(88, 88), rng: random.Random | None = None):
    """frames uint8 [T, H, W] -> (float32 [T, th, tw], (dy, dx, flip))."""
    t, h, w = frames.shape
    th, tw = crop
    x = (frames - 0.0) / 255.0  # Normalize(0.0, 255.0), transform.py:63-64
    if mode == "train":
        rng = rng or random
        dx = rng.randint(0, w - tw)  # RandomCrop, transform.py:120-125
        dy = rng.randint(0, h - th)
        x = x[:, dy:dy + th, dx:dx + tw]
        flip = rng.random() < 0.5  # HorizontalFlip(0.5), transform.py:143-147
        if flip:
            x = x[:, :, ::-1]
    else:
        dx = int(round((w - tw)) / 2.0)  # CenterCrop, transform.py:96-101
        dy = int(round((h - th)) / 2.0)
        x = x[:, dy:dy + th, dx:dx + tw]
        flip = False
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x).astype(np.float32), (dy, dx, int(flip))


def roi_inputs(T: int, H: int, W: int, seed: int) -> np.ndarray:
    return np.random.RandomState(seed).randint(0, 256, size=(T, H, W), dtype=np.uint8)
