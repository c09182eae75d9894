"""Data formats either side of the path: manifests + utterance reader.  The index construction is pinned to what the
reference's AVSpeechDataset.__init__ produced (tests/golden/manifest.json); the mouth stream to the ROI oracle (itself pinned)."""
import contextlib
import io
import json
import os
import random
import struct
import wave

import numpy as np
import pytest
import torch

from oracle.gen_golden_manifest import synthetic_manifests
from oracle.roi_ref import preprocess, roi_inputs
from rtfs_net_amd.datas import AVSpeechDataset, normalize_tensor_wav, read_wav
from tests.util import GOLDEN

GOLD = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.mark.parametrize("n_src", [1, 2])
@pytest.mark.parametrize("segment", [None, 2.0, 2.5])
def test_index_matches_reference(tmp_path, n_src, segment):
    synthetic_manifests(str(tmp_path))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ds = AVSpeechDataset(json_dir=str(tmp_path), n_src=n_src, sample_rate=16000, segment=segment)
    g = GOLD[f"n{n_src}_seg{segment}"]
    assert ds.mix == g["mix"] and ds.sources == g["sources"] and len(ds) == g["len"] and buf.getvalue() == g["printed"]


def test_constructor_errors():
    with pytest.raises(ValueError):
        AVSpeechDataset(json_dir=None)
    with pytest.raises(ValueError):
        AVSpeechDataset(json_dir="/tmp", n_src=3)


def _write_pcm16(path, x):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(16000)
        w.writeframes(x.astype("<i2").tobytes())


def _corpus(d, n=3, L=36000):
    rs = np.random.RandomState(5)
    mix, s1, s2 = [], [], []
    for i in range(n):
        wavs = [rs.randint(-20000, 20000, size=L + 100 * i).astype(np.int16) for _ in range(3)]
        for k, name in enumerate(("mix", "s1", "s2")):
            os.makedirs(os.path.join(d, name), exist_ok=True)
            _write_pcm16(os.path.join(d, name, f"u{i}.wav"), wavs[k])
        mix.append([os.path.join(d, "mix", f"u{i}.wav"), L + 100 * i])
        for k, lst in ((1, s1), (2, s2)):
            npz = os.path.join(d, f"m{k}_{i}.npz")
            np.savez(npz, data=roi_inputs(50, 96, 96, 100 * k + i))
            lst.append([os.path.join(d, f"s{k}", f"u{i}.wav"), npz, L + 100 * i])
    for name, obj in (("mix", mix), ("s1", s1), ("s2", s2)):
        json.dump(obj, open(os.path.join(d, name + ".json"), "w"))
    return mix, s1, s2


def test_getitem_test_mode_single_source(tmp_path):
    d = str(tmp_path)
    mix, s1, s2 = _corpus(d)
    ds = AVSpeechDataset(json_dir=d, n_src=1, sample_rate=16000, segment=None, normalize_audio=True, return_src_path=True)
    assert len(ds) == 6
    mixture, source, mouth, name, src_path = ds[3]  # utterance 1, speaker 2
    raw_mix = np.frombuffer(open(mix[1][0], "rb").read()[44:], dtype="<i2").astype(np.float32) / 32768.0
    raw_src = np.frombuffer(open(s2[1][0], "rb").read()[44:], dtype="<i2").astype(np.float32) / 32768.0
    m, s = torch.from_numpy(raw_mix), torch.from_numpy(raw_src)
    std = m.std(-1, keepdim=True)
    assert torch.equal(mixture, normalize_tensor_wav(m, std=std)[:32000]) and torch.equal(source, normalize_tensor_wav(s, std=std)[:32000])
    want, _ = preprocess(roi_inputs(50, 96, 96, 201), "val")
    assert mouth.shape == (1, 50, 88, 88) and np.array_equal(mouth[0].numpy(), want)
    assert name == "u1.wav" and src_path == s2[1][0]


def test_getitem_train_mode_two_sources_raw_mouth(tmp_path):
    d = str(tmp_path)
    mix, s1, s2 = _corpus(d)
    ds = AVSpeechDataset(json_dir=d, n_src=2, sample_rate=16000, segment=2.0, raw_mouth=True)
    random.seed(77)
    mixture, sources, rois, crops, name = ds[0]  # training order is back to front: utterance 2
    assert name == "u2.wav" and mixture.shape == (32000,) and sources.shape == (2, 32000)
    assert rois.dtype == torch.uint8 and rois.shape == (2, 50, 96, 96) and crops.shape == (2, 3)
    rng = random.Random(77)
    for k in range(2):
        frames = roi_inputs(50, 96, 96, 100 * (k + 1) + 2)
        assert np.array_equal(rois[k].numpy(), frames)
        _, crop = preprocess(frames, "train", rng=rng)
        assert tuple(crops[k].tolist()) == crop
    # host float path draws the same crops and equals the oracle
    ds2 = AVSpeechDataset(json_dir=d, n_src=2, sample_rate=16000, segment=2.0)
    random.seed(77)
    mouth = ds2[0][2]
    rng = random.Random(77)
    for k in range(2):
        want, _ = preprocess(roi_inputs(50, 96, 96, 100 * (k + 1) + 2), "train", rng=rng)
        assert np.array_equal(mouth[k].numpy(), want)


def test_read_wav_encodings(tmp_path):
    x = np.linspace(-0.9, 0.9, 1000).astype(np.float32)
    for code, bits, payload, want in ((3, 32, x.astype("<f4").tobytes(), x),
                                      (1, 32, (x.astype(np.float64) * 2147483648.0).astype("<i4").tobytes(), None)):
        p = tmp_path / f"e{code}{bits}.wav"
        hdr = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, code, 1, 16000, 16000 * bits // 8, bits // 8, bits)
        p.write_bytes(hdr + b"data" + struct.pack("<I", len(payload)) + payload)
        y = read_wav(str(p), 10, 500)
        assert y.dtype == np.float32 and y.shape == (490,)
        assert np.allclose(y, x[10:500], atol=1e-6) if want is None else np.array_equal(y, want[10:500])
    (tmp_path / "bad.wav").write_bytes(b"nope")
    with pytest.raises(ValueError):
        read_wav(str(tmp_path / "bad.wav"))
