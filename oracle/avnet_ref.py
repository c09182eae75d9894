"""ORACLE (test infrastructure, not product code): functional CPU restatement of RTFS-Net's
separation forward path, `AVNet.forward` (/root/reference/src/models/tdavnet.py:86-97).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
It runs on reference-keyed state dicts in the reference's own NCHW layout with
plain torch CPU ops; every function cites the reference lines it follows.  It is pinned against
outputs of the imported reference itself (tests/golden/*.npz, made by oracle/gen_golden.py) --
except for the SRU arithmetic, which lives in an absent third-party package: see oracle/sru_ref.py
("parity unpinned" for that one function).

Scope: the RTFS-Net configuration family (config/*_RTFSNet_*_layer.yaml): STFTEncoder/STFTDecoder,
TDANet separators (2-D audio with DualPathRNN(SRU)+MultiHeadSelfAttention2D, 1-D video with
GlobalAttention), ATTNFusion, MaskGenerator(RI_split).  Anything else raises ValueError.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .sru_ref import sru_forward

EPS = 1e-5  # src/models/layers/normalizations.py:5


# ------------------------------------------------------------------------------------------------
# parameter access
# ------------------------------------------------------------------------------------------------
class P:
    """Prefix view over a flat state dict."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def sub(self, name):
        return P(self.sd, f"{self.prefix}{name}.")

    def __getitem__(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd


# ------------------------------------------------------------------------------------------------
# building blocks (a13)
# ------------------------------------------------------------------------------------------------
def gln(x, w, b):
    """GlobalLayerNorm = GroupNorm(1, C): normalizations.py:8-17."""
    return F.group_norm(x, 1, w, b, EPS)


def ln4d(x, gamma, beta):
    """LayerNormalization4D: normalizations.py:20-37 (biased variance, eps inside the sqrt)."""
    dim = (1, 3) if gamma.shape[-1] > 1 else (1,)
    mu = x.mean(dim=dim, keepdim=True)
    std = torch.sqrt(x.var(dim=dim, unbiased=False, keepdim=True) + EPS)
    return (x - mu) / std * gamma + beta


def batchnorm(x, p: P, training=False):
    """nn.BatchNorm{1,2}d in eval mode (running statistics) or train mode (batch statistics)."""
    return F.batch_norm(x, p["running_mean"], p["running_var"], p["weight"], p["bias"], training, 0.1, EPS)


def _norm(x, p: P, kind, training=False):
    if kind is None:
        return x
    if kind == "gLN":
        q = p.sub("norm")
        return gln(x, q["weight"], q["bias"])
    if kind in ("BatchNorm1d", "BatchNorm2d"):
        return batchnorm(x, p, training)
    raise ValueError(f"unsupported norm {kind}")


ACT_PROBE = None  # tests only: callable(x, kind) that sees the input of every PReLU / ReLU (tests/util.py: kink margins of the gradient checks)


def _act(x, p: P | None, kind):
    if kind is None:
        return x
    if ACT_PROBE is not None and kind in ("PReLU", "ReLU"):
        ACT_PROBE(x, kind)
    if kind == "PReLU":
        return F.prelu(x, p["weight"])
    if kind == "ReLU":
        return F.relu(x)
    if kind == "Sigmoid":
        return torch.sigmoid(x)
    raise ValueError(f"unsupported act {kind}")


def conv_norm_act(x, p: P, *, is2d, stride=1, groups=1, pre_norm=None, pre_act=None, norm=None, act=None, training=False):
    """ConvNormAct: conv_layers.py:65-129.  Sequential index: 0 pre_norm, 1 pre_act, 2 conv, 3 norm, 4 act.
    Padding rule conv_layers.py:100-101: 'same' when stride == 1 (for even kernels torch pads
    left (k-1)//2, right k-1-(k-1)//2), else (k-1)//2 on both sides."""
    fl = p.sub("full_layer")
    x = _norm(x, fl.sub("0"), pre_norm, training)
    x = _act(x, fl.sub("1"), pre_act)
    w = fl["2.weight"]
    b = fl["2.bias"] if fl.has("2.bias") else None
    k = w.shape[-1]
    pad = "same" if stride == 1 else (k - 1) // 2
    conv = F.conv2d if is2d else F.conv1d
    x = conv(x, w, b, stride=stride, padding=pad, groups=groups)
    x = _norm(x, fl.sub("3"), norm, training)
    x = _act(x, fl.sub("4"), act)
    return x


def conv_act_norm_ln4d(x, p: P):
    """ConvActNorm with PReLU + LayerNormalization4D: conv_layers.py:142-205."""
    x = F.conv2d(x, p["conv.weight"], p["conv.bias"])
    x = F.prelu(x, p["act.weight"])
    return ln4d(x, p["norm.gamma"], p["norm.beta"])


# ------------------------------------------------------------------------------------------------
# a1 / a12: STFT encoder and iSTFT decoder
# ------------------------------------------------------------------------------------------------
def to_2d(x):
    """BaseEncoder.unsqueeze_to_2D: encoder.py:18-25."""
    if x.ndim == 1:
        return x.reshape(1, -1)
    if x.ndim == 3:
        assert x.shape[1] == 1
        return x.reshape(x.shape[0], -1)
    return x


def stft_frames(x, win, hop):
    """torch.stft(center=True, reflect, periodic hann, onesided) then stack/transpose: encoder.py:161-173.
    [B, L] -> [B, 2, T, F]"""
    window = torch.hann_window(win).to(x.dtype)  # (the reference's buffer is built in float32, encoder.py:159, and only cast by .double())
    spec = torch.stft(x, n_fft=win, hop_length=hop, window=window, return_complex=True)
    return torch.stack([spec.real, spec.imag], 1).transpose(2, 3).contiguous()


def encoder(x, p: P, cfg):
    spec = stft_frames(to_2d(x), cfg["win"], cfg["hop"])
    w = p["conv.full_layer.2.weight"]
    b = p["conv.full_layer.2.bias"] if p.has("conv.full_layer.2.bias") else None
    return F.conv2d(spec, w, b, padding="same")  # norm_type/act_type are None in RTFS configs (yaml:31-32)


def decoder(x, length, p: P, cfg):
    """STFTDecoder.forward: decoder.py:110-132. x: [B, n_src, C, T, F] -> [B, n_src, L]"""
    B, n_src = x.shape[0], x.shape[1]
    x = x.reshape(B * n_src, x.shape[2], *x.shape[-2:])
    k = p["decoder.weight"].shape[-1]
    y = F.conv_transpose2d(x, p["decoder.weight"], p["decoder.bias"] if p.has("decoder.bias") else None, padding=(k - 1) // 2)
    spec = torch.complex(y[:, 0], y[:, 1]).transpose(1, 2).contiguous()
    window = torch.hann_window(cfg["win"]).to(x.dtype)  # (float32-built buffer, decoder.py:108)
    out = torch.istft(spec, n_fft=cfg["win"], hop_length=cfg["hop"], window=window, length=length)
    return out.view(B, n_src, length)


# ------------------------------------------------------------------------------------------------
# a6 / a7: dual-path SRU
# ------------------------------------------------------------------------------------------------
def dual_path_rnn(x, p: P, *, dim, hid, ksize=8, stride=1, num_layers=4):
    """DualPathRNN.forward: rnn_layers.py:136-162.  dim=4 runs along F, dim=3 along T."""
    if dim == 4:
        x = x.transpose(-2, -1).contiguous()
    B, C, oT, oF = x.shape
    nT = math.ceil((oT - ksize) / stride) * stride + ksize
    nF = math.ceil((oF - ksize) / stride) * stride + ksize
    x = F.pad(x, (0, nF - oF, 0, nT - oT))
    residual = x
    x = ln4d(x, p["norm.gamma"], p["norm.beta"])
    x = x.permute(0, 3, 1, 2).contiguous().view(B * nF, C, nT, 1)
    x = F.unfold(x, (ksize, 1), stride=(stride, 1))  # [B*nF, C*ksize, L], feature index c*ksize + k
    x = x.permute(2, 0, 1).contiguous()
    layers = [
        {k: p[f"rnn.rnn_lst.{i}.{k}"] for k in ("weight", "weight_c", "bias", "scale_x")} for i in range(num_layers)
    ]
    x, _ = sru_forward(x, layers, hid, True)
    x = x.permute(1, 2, 0)
    x = F.conv_transpose1d(x, p["linear.weight"], p["linear.bias"], stride=stride)
    x = x.view(B, nF, C, nT).permute(0, 2, 3, 1).contiguous()
    x = (x + residual)[..., :oT, :oF]
    if dim == 4:
        x = x.transpose(-2, -1).contiguous()
    return x


# ------------------------------------------------------------------------------------------------
# a8: TF self-attention
# ------------------------------------------------------------------------------------------------
def mhsa2d(x, p: P, n_head):
    """MultiHeadSelfAttention2D.forward: attention.py:149-189 (dim == 3)."""
    B, C, T, Fq = x.shape
    res = x
    Q = torch.cat([conv_act_norm_ln4d(x, p.sub(f"Queries.{h}")) for h in range(n_head)], 0)
    K = torch.cat([conv_act_norm_ln4d(x, p.sub(f"Keys.{h}")) for h in range(n_head)], 0)
    V = torch.cat([conv_act_norm_ln4d(x, p.sub(f"Values.{h}")) for h in range(n_head)], 0)
    Q = Q.transpose(1, 2).flatten(2)
    K = K.transpose(1, 2).flatten(2)
    V = V.transpose(1, 2)
    vshape = V.shape
    V = V.flatten(2)
    att = torch.softmax(Q @ K.transpose(1, 2) / (Q.shape[-1] ** 0.5), dim=2)
    V = (att @ V).reshape(vshape).transpose(1, 2)
    e = V.shape[1]
    x = V.reshape(n_head, B, e, T, Fq).transpose(0, 1).reshape(B, n_head * e, T, Fq)
    x = conv_act_norm_ln4d(x, p.sub("attn_concat_proj"))
    return x + res


# ------------------------------------------------------------------------------------------------
# a9: video-branch global attention (eval mode: dropout / drop-path are identity)
# ------------------------------------------------------------------------------------------------
def positional_encoding(T, C, max_len=10000):
    """PositionalEncoding buffer: attention.py:9-25."""
    pe = torch.zeros(max_len, C)
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, C, 2).float() * -(torch.log(torch.tensor(max_len).float()) / C))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe[:T].unsqueeze(0)


def global_attention_1d(x, p: P, n_head, ffn_kernel):
    """GlobalAttention = MultiHeadSelfAttention + FeedForwardNetwork: attention.py:28-73,192-220; conv_layers.py:218-259."""
    m = p.sub("MHSA")
    res = x
    y = x.transpose(1, 2)
    C = y.shape[-1]
    y = F.layer_norm(y, (C,), m["norm1.weight"], m["norm1.bias"], EPS)
    y = y + (m["pos_enc.pe"][:, : y.shape[1]] if m.has("pos_enc.pe") else positional_encoding(y.shape[1], C))
    r2 = y
    y, _ = F.multi_head_attention_forward(
        y.transpose(0, 1), y.transpose(0, 1), y.transpose(0, 1), C, n_head,
        m["attention.in_proj_weight"], m["attention.in_proj_bias"], None, None, False, 0.0,
        m["attention.out_proj.weight"], m["attention.out_proj.bias"], training=False, need_weights=False,
    )
    y = y.transpose(0, 1) + r2
    y = F.layer_norm(y, (C,), m["norm2.weight"], m["norm2.bias"], EPS)
    x = y.transpose(2, 1) + res
    f = p.sub("FFN")
    res = x
    y = conv_norm_act(x, f.sub("encoder"), is2d=False, norm="gLN")
    hidc = y.shape[1]
    y = conv_norm_act(y, f.sub("refiner"), is2d=False, groups=hidc, act="ReLU")
    y = conv_norm_act(y, f.sub("decoder"), is2d=False, norm="gLN")
    return y + res


# ------------------------------------------------------------------------------------------------
# a5.6: TFAR unit
# ------------------------------------------------------------------------------------------------
def injection_multi_sum(local, glob, p: P, *, is2d, norm, training=False):
    """InjectionMultiSum.forward: layers/fusion.py:54-69."""
    nd = local.ndim // 2
    old, new = glob.shape[-nd:], local.shape[-nd:]
    H = local.shape[1]
    kw = dict(is2d=is2d, groups=H, norm=norm, training=training)
    local_emb = conv_norm_act(local, p.sub("local_embedding"), **kw)
    if math.prod(new) > math.prod(old):
        g_emb = F.interpolate(conv_norm_act(glob, p.sub("global_embedding"), **kw), size=new, mode="nearest")
        gate = F.interpolate(conv_norm_act(glob, p.sub("global_gate"), act="Sigmoid", **kw), size=new, mode="nearest")
    else:
        gi = F.interpolate(glob, size=new, mode="nearest")
        g_emb = conv_norm_act(gi, p.sub("global_embedding"), **kw)
        gate = conv_norm_act(gi, p.sub("global_gate"), act="Sigmoid", **kw)
    return local_emb * gate + g_emb


# ------------------------------------------------------------------------------------------------
# a5 / a9: RTFS block and VP block (same class in the reference: separators/tdanet.py:8-133)
# ------------------------------------------------------------------------------------------------
def tdanet_block(x, p: P, net, *, training=False, taps=None):
    """TDANetBlock.forward: separators/tdanet.py:106-133. `net` = the audio_params / video_params dict."""
    is2d = net.get("is2d", False)
    norm, act = net.get("norm_type", "gLN"), net.get("act_type", "PReLU")
    depth, stride, H = net.get("upsampling_depth", 4), net.get("stride", 2), net["hid_chan"]
    C = x.shape[1]
    residual = conv_norm_act(x, p.sub("gateway"), is2d=is2d, groups=C, act=act)
    x_enc = conv_norm_act(residual, p.sub("projection"), is2d=is2d, norm=norm, act=act, training=training)
    ds = [conv_norm_act(x_enc, p.sub("downsample_layers.0"), is2d=is2d, groups=H, norm=norm, training=training)]
    for i in range(1, depth):
        ds.append(conv_norm_act(ds[-1], p.sub(f"downsample_layers.{i}"), is2d=is2d, stride=stride, groups=H, norm=norm, training=training))
    nd = ds[-1].ndim // 2
    size = ds[-1].shape[-nd:]
    pool = F.adaptive_avg_pool2d if is2d else F.adaptive_avg_pool1d
    g = sum(pool(f, output_size=size) for f in ds)
    if taps is not None:
        taps["gateway"], taps["proj"], taps["pooled"] = residual, x_enc, g
        for i, d in enumerate(ds):
            taps[f"ds{i}"] = d
    for name, layer in net.get("layers", {}).items():
        idx = list(net["layers"].keys()).index(name)
        q = p.sub(f"globalatt.{idx}")
        lt = layer["layer_type"]
        if lt == "DualPathRNN":
            if layer.get("rnn_type") != "SRU":
                raise ValueError("oracle supports rnn_type SRU only")
            g = dual_path_rnn(g, q, dim=layer["dim"], hid=layer["hid_chan"], ksize=layer.get("kernel_size", 8),
                              stride=layer.get("stride", 1), num_layers=layer.get("num_layers", 1))
        elif lt == "MultiHeadSelfAttention2D":
            g = mhsa2d(g, q, layer.get("n_head", 4))
        elif lt == "GlobalAttention":
            g = global_attention_1d(g, q, layer.get("n_head", 8), layer.get("kernel_size", 5))
        else:
            raise ValueError(f"unsupported layer_type {lt}")
        if taps is not None:
            taps[f"globalatt.{idx}"] = g
    fused = [injection_multi_sum(ds[i], g, p.sub(f"fusion_layers.{i}"), is2d=is2d, norm=norm, training=training) for i in range(depth)]
    exp = injection_multi_sum(fused[-2], fused[-1], p.sub(f"concat_layers.{depth - 2}"), is2d=is2d, norm=norm, training=training) + ds[-2]
    for i in range(depth - 3, -1, -1):
        exp = injection_multi_sum(fused[i], exp, p.sub(f"concat_layers.{i}"), is2d=is2d, norm=norm, training=training) + ds[i]
    if taps is not None:
        taps["expanded"] = exp
        for i, f in enumerate(fused):
            taps[f"fused{i}"] = f
    return conv_norm_act(exp, p.sub("residual_conv"), is2d=is2d) + residual


# ------------------------------------------------------------------------------------------------
# a10: CAF
# ------------------------------------------------------------------------------------------------
def caf(audio, video, p: P, heads, training=False):
    """ATTNFusionCell.forward: layers/fusion.py:252-274 (audio side only: TDAVNet/fusion.py:202,210)."""
    B, C, T, _ = audio.shape
    Cv = video.shape[1]
    resize = conv_norm_act(video, p.sub("resize"), is2d=False, groups=C, norm="gLN")
    bt = F.interpolate(resize, size=T, mode="nearest").unsqueeze(-1)
    k1 = conv_norm_act(audio, p.sub("key_embed"), is2d=True, groups=C, norm="BatchNorm2d", act="ReLU", training=training) * bt
    v = conv_norm_act(audio, p.sub("value_embed"), is2d=True, groups=C, norm="BatchNorm2d", training=training)
    att = conv_norm_act(video, p.sub("attention_embed"), is2d=False, groups=C, norm="gLN")
    att = att.reshape(B, C, heads, -1).mean(2).view(B, C, -1)
    att = F.interpolate(torch.softmax(att, -1), size=T, mode="nearest").unsqueeze(-1)
    return k1 + att * v


# ------------------------------------------------------------------------------------------------
# a11: S3 mask
# ------------------------------------------------------------------------------------------------
def s3_mask(refined, a_emb, p: P, n_src, mask_act="ReLU"):
    """MaskGenerator.forward + __apply_masks with RI_split: mask_generator.py:67-99."""
    B, C = a_emb.shape[0], a_emb.shape[1]
    m = F.prelu(refined, p["mask_generator.0.weight"])
    m = conv_norm_act(m, p.sub("mask_generator.1"), is2d=True, act=mask_act)
    dims = a_emb.shape[-2:]
    m = m.view(B, n_src, 2, C // 2, *dims)
    e = a_emb.view(B, 2, C // 2, *dims)
    mr, mi = m[:, :, 0], m[:, :, 1]
    er, ei = e[:, 0].unsqueeze(1), e[:, 1].unsqueeze(1)
    return torch.cat([er * mr - ei * mi, er * mi + ei * mr], 2)


# ------------------------------------------------------------------------------------------------
# a4 + top level
# ------------------------------------------------------------------------------------------------
def normalise_cfg(audionet: dict) -> dict:
    """Flatten the YAML `audionet:` section (config/lrs2_RTFSNet_4_layer.yaml:8-104) into what the oracle needs."""
    ed = audionet["enc_dec_params"]
    if ed.get("encoder_type") != "STFTEncoder" or ed.get("decoder_type") != "STFTDecoder":
        raise ValueError("oracle supports STFTEncoder/STFTDecoder only")
    if audionet["audio_params"].get("audio_net") != "TDANet" or audionet.get("video_params", {}).get("video_net") != "TDANet":
        raise ValueError("oracle supports TDANet separators only")
    if audionet.get("fusion_params", {}).get("fusion_type") != "ATTNFusion":
        raise ValueError("oracle supports ATTNFusion only")
    mg = audionet["mask_generation_params"]
    if mg.get("mask_generator_type", "MaskGenerator") != "MaskGenerator" or not mg.get("RI_split", False):
        raise ValueError("oracle supports MaskGenerator with RI_split only")
    return dict(
        win=ed["win"], hop=ed["hop_length"], n_src=audionet["n_src"],
        audio=audionet["audio_params"], video=audionet["video_params"],
        heads=audionet["fusion_params"].get("kernel_size", 1), mask_act=mg.get("mask_act", "ReLU"),
        bn=audionet["audio_bn_params"],
    )


def avnet_forward(sd: dict, audionet: dict, wav: torch.Tensor, emb: torch.Tensor, taps: dict | None = None, training=False):
    """AVNet.forward: tdavnet.py:86-97; RefinementModule.forward: refinement_module.py:45-62."""
    cfg = normalise_cfg(audionet)
    root = P(sd)
    L = wav.shape[-1]
    a_emb = encoder(wav, root.sub("encoder"), cfg)
    a0 = conv_norm_act(a_emb, root.sub("audio_bottleneck"), is2d=True,
                       pre_norm=cfg["bn"].get("pre_norm_type"), pre_act=cfg["bn"].get("pre_act_type"))
    video = emb  # video_bottleneck is nn.Identity for kernel_size -1 (tdavnet.py:60; conv_layers.py:118-119)
    rm = root.sub("refinement_module")
    fusion_repeats = cfg["video"].get("repeats", 0)
    R = cfg["audio"]["repeats"]
    if fusion_repeats != 1:
        raise ValueError("oracle supports video repeats == 1 (RTFS-Net configs)")
    if taps is not None:
        taps["a_emb"], taps["a0"] = a_emb, a0
    shared = bool(cfg["audio"].get("shared", False))  # tdanet.py:170-181: one block applied R times, or R blocks (state-dict prefix blocks.<i>.)
    ablk_of = lambda i: rm.sub("audio_net.blocks") if shared else rm.sub(f"audio_net.blocks.{i}")  # noqa: E731
    ablk, vblk = ablk_of(0), rm.sub("video_net.blocks")
    btaps = {} if taps is not None else None
    a = tdanet_block(a0, ablk, cfg["audio"], training=training, taps=btaps)
    if taps is not None:
        taps["block0"] = a
        taps.update({f"block0.{k}": v for k, v in btaps.items()})
    v1 = tdanet_block(video, vblk, cfg["video"], training=training)
    a = caf(a, v1, rm.sub("crossmodal_fusion.fusion_module.audio_lstm"), cfg["heads"], training=training)
    if taps is not None:
        taps["vp"], taps["caf"] = v1, a
    for i in range(1, R):
        btaps = {} if taps is not None else None
        a = tdanet_block(a + a0, ablk_of(i), cfg["audio"], training=training, taps=btaps)
        if taps is not None:
            taps[f"block{i}"] = a
            taps.update({f"block{i}.{k}": v for k, v in btaps.items()})
    sep = s3_mask(a, a_emb, root.sub("mask_generator"), cfg["n_src"], cfg["mask_act"])
    if taps is not None:
        taps["masked"] = sep
    return decoder(sep, L, root.sub("decoder"), cfg)


# ------------------------------------------------------------------------------------------------
# f1 helper used by parity tests / bench: SI-SDR as in losses/matrix.py:22-53 (sisdr, zero-mean)
# ------------------------------------------------------------------------------------------------
def si_sdr(est: torch.Tensor, target: torch.Tensor, eps=1e-8) -> torch.Tensor:
    est = est - est.mean(-1, keepdim=True)
    target = target - target.mean(-1, keepdim=True)
    dot = (est * target).sum(-1, keepdim=True)
    proj = dot * target / ((target**2).sum(-1, keepdim=True) + eps)
    noise = est - proj
    return 10 * torch.log10((proj**2).sum(-1) / ((noise**2).sum(-1) + eps) + eps)
