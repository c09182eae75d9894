"""Shared helpers for the parity tests."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import synth  # noqa: E402


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    """relative L2 error ||a-b|| / ||b||"""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cl_to_nchw(t: torch.Tensor, B, T, F, C):
    """channels-last device buffer -> [B, C, T, F]"""
    return t.view(B, T, F, C).permute(0, 3, 1, 2).contiguous()


def nchw_to_cl(t: torch.Tensor):
    return t.permute(0, 2, 3, 1).contiguous()


def make_model(repeats: int, device="cpu", salt=0):
    """Product AVNet with the deterministic synthetic weights of oracle/synth.py; returns (model, state_dict_cpu, audionet)."""
    from rtfs_net_amd import AVNet

    cfg = synth.rtfs_audionet(repeats)
    model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
    sd = synth.synth_state_dict(model.state_dict(), salt)
    model.load_state_dict(sd)
    return model.to(device), sd, cfg


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


from oracle.regimes import VIDEO_PREFIX, smooth_regime, stable_emb  # noqa: E402,F401  (oracle-side: kink-stable inputs, smooth-regime weights)
