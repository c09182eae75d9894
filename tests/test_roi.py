"""Mouth-ROI preprocessing: oracle vs vectors produced by the reference's transform classes (CPU); the HIP kernel
bit-exact against the oracle (GPU) - byte/index work plus one table lookup, so the bar is equality, not a tolerance."""
import random
import zlib

import numpy as np
import pytest
import torch

from oracle.roi_ref import preprocess, roi_inputs
from rtfs_net_amd.models import videomodels
from tests.util import load_npz

GOLD = load_npz("roi.npz")
CASES = ["val96", "val_odd", "val88", "train96", "train_b", "train_odd"]


def _case(name):
    T, H, W, seed, train = (int(v) for v in GOLD[f"{name}_cfg"])
    return roi_inputs(T, H, W, seed), "train" if train else "val", seed


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    frames, mode, seed = _case(name)
    y, crop = preprocess(frames, mode, rng=random.Random(seed))
    assert tuple(crop) == tuple(int(v) for v in GOLD[f"{name}_crop"])
    assert zlib.crc32(y.tobytes()) == int(GOLD[f"{name}_crc"][0])
    assert np.array_equal(y[:, ::9, ::7], GOLD[f"{name}_sample"])


def test_host_side_draws_and_errors():
    r = videomodels.MouthROI()
    for name in CASES:
        T, H, W, seed, train = (int(v) for v in GOLD[f"{name}_cfg"])
        want = tuple(int(v) for v in GOLD[f"{name}_crop"])
        if train:  # same draws, in the same order, as RandomCrop + HorizontalFlip
            assert tuple(r.random_crops(1, H, W, rng=random.Random(seed))[0].tolist()) == want
        else:
            assert r.center_offsets(H, W) == want[:2]
    with pytest.raises(RuntimeError):  # no CPU fallback
        r(torch.zeros(1, 2, 96, 96, dtype=torch.uint8))
    with pytest.raises(ValueError):
        r.padded(torch.zeros(1, 2, 96, 96))


@pytest.mark.gpu
@pytest.mark.parametrize("names", [["val96"], ["val_odd"], ["val88"], ["train96", "train_b"], ["train_odd"]])
def test_hip_bit_exact(names):
    r = videomodels.MouthROI()
    clips, crops, want, train = [], [], [], False
    for n in names:
        frames, mode, seed = _case(n)
        y, crop = preprocess(frames, mode, rng=random.Random(seed))
        clips.append(torch.from_numpy(frames)), crops.append(crop), want.append(y)
        train = mode == "train"
    rois = torch.stack(clips).cuda()
    ct = torch.tensor(crops, dtype=torch.int32) if train else None
    out = r(rois, ct)
    assert out.shape == (len(names), 1, rois.shape[1], 88, 88)
    assert np.array_equal(out[:, 0].cpu().numpy(), np.stack(want))
    P = r.padded(rois, ct).cpu()
    assert float(P[:, :2].abs().sum() + P[:, -2:].abs().sum() + P[:, :, :3].abs().sum() + P[:, :, -3:].abs().sum()
                 + P[:, :, :, :3].abs().sum() + P[:, :, :, -3:].abs().sum()) == 0.0
    if not train:  # centre crop given explicitly == default
        dy, dx = r.center_offsets(rois.shape[2], rois.shape[3])
        assert torch.equal(r(rois, torch.tensor([[dy, dx, 0]] * len(names), dtype=torch.int32)), out)
    with pytest.raises(ValueError):
        r(rois, torch.tensor([[0, rois.shape[3], 0]] * len(names), dtype=torch.int32))


@pytest.mark.gpu
def test_forward_rois_equals_forward_of_preprocessed():
    from oracle import synth

    m = videomodels.FRCNNVideoModel(print_macs=False)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), salt=3))
    m = m.cuda()
    m.eval()
    rois = torch.from_numpy(np.stack([roi_inputs(4, 96, 96, 31), roi_inputs(4, 96, 96, 32)])).cuda()
    crops = torch.tensor([[3, 7, 1], [8, 0, 0]], dtype=torch.int32)
    with torch.no_grad():
        a = m.forward_rois(rois, crops)
        b = m(videomodels.MouthROI()(rois, crops))
    assert a.shape == (2, 512, 4) and torch.equal(a, b)
