"""Deterministic synthetic weights and inputs (no arithmetic of the path lives here).

bench.py and `__graft_entry__.smoke()` feed the product from these generators; the oracle side (oracle/synth.py re-exports this
module) uses the SAME functions in oracle/gen_golden.py, which loads them into the imported reference, and in the parity tests -
so the reference, the oracle and the HIP path all see identical numbers without any checkpoint or dataset (there is no network).
Inputs follow SURVEY.md §8d.
"""
from __future__ import annotations

import copy
import zlib

import torch

INPUT_SEED = 20240229

# A reduced member of the RTFS-Net family (same operator graph as config/lrs2_RTFSNet_4_layer.yaml:8-104,
# smaller channel counts / window) used for stage-level fixtures that must stay small in git.
TINY_AUDIONET = {
    "n_src": 1,
    "pretrained_vout_chan": 32,
    "video_bn_params": {"kernel_size": -1},
    "audio_bn_params": {"pre_norm_type": "gLN", "pre_act_type": "ReLU", "out_chan": 32, "kernel_size": 1, "is2d": True},
    "enc_dec_params": {"encoder_type": "STFTEncoder", "decoder_type": "STFTDecoder", "win": 64, "hop_length": 32,
                       "out_chan": 32, "kernel_size": 3, "stride": 1, "bias": False, "act_type": None, "norm_type": None},
    "audio_params": {
        "audio_net": "TDANet", "hid_chan": 16, "kernel_size": 4, "stride": 2, "norm_type": "gLN", "act_type": "PReLU",
        "upsampling_depth": 2, "repeats": 2, "shared": True, "is2d": True,
        "layers": {
            "layer_1": {"layer_type": "DualPathRNN", "hid_chan": 8, "dim": 4, "kernel_size": 8, "stride": 1,
                        "rnn_type": "SRU", "num_layers": 4, "bidirectional": True},
            "layer_2": {"layer_type": "DualPathRNN", "hid_chan": 8, "dim": 3, "kernel_size": 8, "stride": 1,
                        "rnn_type": "SRU", "num_layers": 4, "bidirectional": True},
            "layer_3": {"layer_type": "MultiHeadSelfAttention2D", "dim": 3, "n_freqs": 16, "n_head": 4, "hid_chan": 4,
                        "act_type": "PReLU", "norm_type": "LayerNormalization4D"},
        },
    },
    "video_params": {
        "video_net": "TDANet", "hid_chan": 16, "kernel_size": 3, "stride": 2, "norm_type": "BatchNorm1d", "act_type": "PReLU",
        "upsampling_depth": 4, "repeats": 1, "shared": True, "is2d": False,
        "layers": {"layer_1": {"layer_type": "GlobalAttention", "ffn_name": "FeedForwardNetwork", "kernel_size": 3,
                               "n_head": 8, "dropout": 0.1}},
    },
    "fusion_params": {"fusion_type": "ATTNFusion", "fusion_shared": True, "kernel_size": 4, "is2d": True},
    "mask_generation_params": {"mask_generator_type": "MaskGenerator", "mask_act": "ReLU", "RI_split": True, "is2d": True},
}


def rtfs_audionet(repeats: int = 4) -> dict:
    """The `audionet:` section of config/lrs2_RTFSNet_{4,6,12}_layer.yaml (they differ only in repeats, yaml:43)."""
    cfg = copy.deepcopy(TINY_AUDIONET)
    cfg["pretrained_vout_chan"] = 512
    cfg["audio_bn_params"]["out_chan"] = 256
    cfg["enc_dec_params"].update(win=256, hop_length=128, out_chan=256)
    cfg["audio_params"].update(hid_chan=64, repeats=repeats)
    for k in ("layer_1", "layer_2"):
        cfg["audio_params"]["layers"][k]["hid_chan"] = 32
    cfg["audio_params"]["layers"]["layer_3"]["n_freqs"] = 64
    cfg["video_params"]["hid_chan"] = 64
    return cfg


def _gen(key: str, salt: int) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def synth_state_dict(template: dict, salt: int = 0) -> dict:
    """Deterministic 'trained-looking' values for every tensor of a reference-keyed state dict.

    `template` supplies keys, shapes and dtypes (any AVNet state_dict).  Each tensor gets its own
    generator seeded from crc32(key), so the result does not depend on dict order."""
    out = {}
    for k, v in template.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf in ("num_batches_tracked", "scale_x", "pe"):
            out[k] = v.clone()
            continue
        g = _gen(k, salt)
        shp = tuple(v.shape)
        u = lambda lo, hi: torch.rand(shp, generator=g) * (hi - lo) + lo  # noqa: E731
        n = lambda s: torch.randn(shp, generator=g) * s  # noqa: E731
        if leaf == "running_var":
            t = u(0.5, 1.5)
        elif leaf == "running_mean":
            t = n(0.2)
        elif leaf in ("bias", "beta", "in_proj_bias"):
            t = n(0.1)
        elif leaf == "gamma":
            t = u(0.6, 1.4)
        elif leaf == "weight_c":
            t = u(-1.2, 1.2)
        elif leaf == "weight" and v.numel() == 1:
            t = u(0.1, 0.4)  # PReLU slope
        elif leaf == "weight" and v.ndim == 1:
            t = u(0.6, 1.4)  # GroupNorm / BatchNorm / LayerNorm scale
        else:
            if ".rnn_lst." in k:  # SRU weight is [d_in, d_out]
                fan_in = v.shape[0]
            else:  # conv [out, in/groups, k...], linear [out, in]
                fan_in = max(1, v.numel() // v.shape[0])
            b = (3.0 / fan_in) ** 0.5
            t = u(-b, b)
        out[k] = t.to(v.dtype)
    return out


def synth_inputs(B: int, L: int, Tv: int, vdim: int = 512, seed: int = INPUT_SEED):
    """SURVEY.md §8d synthetic inputs: two low-passed Gaussian 'speakers' mixed and clipped + N(0,1) lip embeddings.
    Returns (mix [B, L], s1 [B, L], emb [B, vdim, Tv])."""
    g = torch.Generator()
    g.manual_seed(seed)
    s = 0.1 * torch.randn(2, B, L + 4, generator=g)
    s = s.unfold(-1, 5, 1).mean(-1)  # 5-tap box filter
    mix = (s[0] + s[1]).clamp(-1, 1)
    emb = torch.randn(B, vdim, Tv, generator=g)
    return mix.contiguous(), s[0].contiguous(), emb


def lip_inputs(B: int, T: int, H: int = 88, W: int = 88, seed: int = 20240229) -> torch.Tensor:
    """Synthetic normalised mouth crops: the reference normalises uint8 grey frames with mean 0.421, std 0.165
    (src/datas/transform.py:151-167) -> values in about [-2.6, 3.5]; uniform over that range, fixed seed."""
    g = torch.Generator()
    g.manual_seed(seed + 17 * B + T)
    return (torch.rand(B, 1, T, H, W, generator=g) - 0.421) / 0.165
