"""Dataset manifests and the utterance reader of the reference, for the accelerated path.

Wire formats (written by data-preprocess/preprocess_lrs2.py:44-58, read by src/datas/avspeech_dataset.py:46-54,121-124):
  mix.json          [[wav_path, n_samples], ...]
  s1.json, s2.json  [[wav_path, mouth_npz_path, n_samples], ...]      mouth npz: key "data", uint8 [Tv, H, W] grey ROI frames
`AVSpeechDataset` keeps the reference's constructor, index construction (including its drop rule, its reversed order in training
mode and `len()` counted before the drop) and return tuples.  Difference by design: with `raw_mouth=True` the mouth stream is returned
as the uint8 ROI frames plus the (dy, dx, flip) crop triple instead of a float clip preprocessed on the host - the normalise / crop /
flip arithmetic then runs on the GPU, fused into the lip encoder's input (`videomodels.MouthROI`, `FRCNNVideoModel.forward_rois`),
bit-identical to the host pipeline.  With `raw_mouth=False` the float clip is produced on the host through the same value table.
Audio is read with `soundfile` when it is installed, else with the built-in PCM reader below (16/32-bit PCM and float32 WAV).
"""
from __future__ import annotations

import json
import os
import random
import struct

import numpy as np
import torch
from torch.utils.data import Dataset


def normalize_tensor_wav(wav_tensor, eps=1e-8, std=None):
    """(x - mean) / (std + eps) over the last axis (avspeech_dataset.py:11-15)."""
    mean = wav_tensor.mean(-1, keepdim=True)
    if std is None:
        std = wav_tensor.std(-1, keepdim=True)
    return (wav_tensor - mean) / (std + eps)


def read_wav(path: str, start: int = 0, stop: int | None = None) -> np.ndarray:
    """float32 samples [start:stop] of a mono WAV file, scaled like `soundfile.read(..., dtype="float32")` (PCM16 / 32768, PCM32 / 2^31)."""
    try:
        import soundfile as sf

        if hasattr(sf, "read"):
            return sf.read(path, start=start, stop=stop, dtype="float32")[0]
    except ImportError:
        pass
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", data[pos + 8:pos + 24])
        elif tag == b"data":
            body = data[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    if fmt is None or body is None:
        raise ValueError(f"{path}: missing fmt / data chunk")
    code, channels, _, _, _, bits = fmt
    if code == 1 and bits == 16:
        x = np.frombuffer(body, dtype="<i2").astype(np.float32) / 32768.0
    elif code == 1 and bits == 32:
        x = (np.frombuffer(body, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif code == 3 and bits == 32:
        x = np.frombuffer(body, dtype="<f4").copy()
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format {code}, {bits} bit)")
    if channels > 1:
        x = x.reshape(-1, channels)
    return x[start:stop]


def read_manifests(json_dir: str, n_src: int, seg_len: int | None):
    """-> (mix, sources, length) exactly as AVSpeechDataset.__init__ builds them (avspeech_dataset.py:46-110); seg_len None = test mode."""
    if json_dir is None:
        raise ValueError("JSON DIR is None!")
    if n_src not in [1, 2]:
        raise ValueError("{} is not in [1, 2]".format(n_src))
    with open(os.path.join(json_dir, "mix.json"), "r") as f:
        mix_infos = json.load(f)
    sources_infos = []
    for name in ("s1", "s2"):
        with open(os.path.join(json_dir, name + ".json"), "r") as f:
            sources_infos.append(json.load(f))
    test = seg_len is None
    length = len(mix_infos) * (2 if n_src == 1 else 1)  # counted before the drop, as the reference does
    mix, sources, drop_utt, drop_len = [], [], 0, 0
    if test:
        if n_src == 1:
            for i in range(len(mix_infos)):
                for src in sources_infos:
                    mix.append(mix_infos[i])
                    sources.append(src[i])
        else:
            mix, sources = mix_infos, sources_infos
    else:
        for i in range(len(mix_infos) - 1, -1, -1):  # back to front
            if mix_infos[i][1] < seg_len:
                drop_utt, drop_len = drop_utt + 1, drop_len + mix_infos[i][1]
                del mix_infos[i]
                for src in sources_infos:
                    del src[i]
            elif n_src == 1:
                for src in sources_infos:
                    mix.append(mix_infos[i])
                    sources.append(src[i])
            else:
                mix.append(mix_infos[i])
                sources.append([src[i] for src in sources_infos])
    return mix, sources, length, drop_utt, drop_len


class AVSpeechDataset(Dataset):
    def __init__(self, json_dir: str = "", n_src: int = 2, sample_rate: int = 8000, segment: float = 4.0, normalize_audio: bool = False,
                 return_src_path: bool = False, audio_only: bool = False, raw_mouth: bool = False):
        super().__init__()
        self.json_dir = json_dir
        self.sample_rate = sample_rate
        self.normalize_audio = normalize_audio
        self.return_src_path = return_src_path
        self.audio_only = audio_only
        self.raw_mouth = raw_mouth
        self.seg_len = None if segment is None else int(segment * sample_rate)
        self.n_src = n_src
        self.test = self.seg_len is None
        self.mix, self.sources, self.length, drop_utt, drop_len = read_manifests(json_dir, n_src, self.seg_len)
        if drop_utt > 0:
            print("Drop {} utts({:.2f} h) from {} (shorter than {} samples)".format(drop_utt, drop_len / sample_rate / 3600, self.length,
                                                                                     self.seg_len))
        from ..models.videomodels.roi import MouthROI  # value table + crop rules shared with the GPU kernel

        self._roi = MouthROI()

    def __len__(self):
        return self.length

    def _mouth(self, npz_path: str):
        frames = np.load(npz_path)["data"]  # uint8 [Tv, H, W]
        _, H, W = frames.shape
        if self.test:
            dy, dx = self._roi.center_offsets(H, W)
            flip = 0
        else:
            dy, dx, flip = self._roi.random_crops(1, H, W, rng=random)[0].tolist()
        if self.raw_mouth:
            return torch.from_numpy(frames), torch.tensor([dy, dx, flip], dtype=torch.int32)
        th, tw = self._roi.crop_size
        crop = frames[:, dy:dy + th, dx:dx + tw]
        if flip:
            crop = crop[:, :, ::-1]
        return self._roi._lut_host[torch.from_numpy(np.ascontiguousarray(crop)).long()], None  # table lookup == the float64 pipeline, cast to f32

    def __getitem__(self, idx: int):
        self.EPS = 1e-8
        stop = self.seg_len
        two = self.sample_rate * 2
        if self.n_src == 1:
            mixture = torch.from_numpy(np.ascontiguousarray(read_wav(self.mix[idx][0], 0, stop)))
            source = torch.from_numpy(np.ascontiguousarray(read_wav(self.sources[idx][0], 0, stop)))
            mouths = None if self.audio_only else [self._mouth(self.sources[idx][1])]
            if self.normalize_audio:
                m_std = mixture.std(-1, keepdim=True)
                mixture = normalize_tensor_wav(mixture, eps=self.EPS, std=m_std)
                source = normalize_tensor_wav(source, eps=self.EPS, std=m_std)
            ret = (mixture[:two], source[:two])
            src_path = self.sources[idx][0]
        else:
            mixture = torch.from_numpy(np.ascontiguousarray(read_wav(self.mix[idx][0], 0, stop)))
            source = torch.stack([torch.from_numpy(np.ascontiguousarray(read_wav(src[0], 0, stop))) for src in self.sources[idx]])
            mouths = None if self.audio_only else [self._mouth(src[1]) for src in self.sources[idx]]
            if self.normalize_audio:
                m_std = mixture.std(-1, keepdim=True)
                mixture = normalize_tensor_wav(mixture, eps=self.EPS, std=m_std)
                source = normalize_tensor_wav(source, eps=self.EPS, std=m_std)
            ret = (mixture[:two], source[:two])
            src_path = None
        if mouths is not None:
            ret += (torch.stack([m for m, _ in mouths]),)
            if self.raw_mouth:
                ret += (torch.stack([c for _, c in mouths]),)
        ret += (self.mix[idx][0].split("/")[-1],)
        if self.return_src_path and src_path is not None:
            ret += (src_path,)
        return ret
