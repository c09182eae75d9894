"""Parameter tree of RTFS-Net with the reference's state-dict names and shapes.

The audio branch's INNER modules are PARAMETER HOLDERS: their arithmetic runs fused in the HIP
kernels (hip_path.py), so calling one of them directly raises.  What the reference's module protocol
offers above them works (models/stage_views.py): the five stage modules of AVNet are callable one by
one (NCHW in / out), and forward hooks on an RTFS block, its direct children, the VP block and the
CAF cell fire with the module's inputs and output.  Only the tiny video-branch (VP) block -- 50
tokens, ~0.05 % of the MACs, SURVEY.md §2 row 9 / §8 a9 -- also has a torch `forward` (PyTorch-ROCm
glue: the training step on fewer than 8 video frames, `RTFS_DISABLE=vp_hip`).

Naming follows the reference so that checkpoints load unchanged:
  ConvNormAct.full_layer.{0 pre_norm,1 pre_act,2 conv,3 norm,4 act}   src/models/layers/conv_layers.py:121-127
  ConvActNorm.{conv,act,norm}                                         conv_layers.py:184-205
  GlobalLayerNorm.norm (GroupNorm(1,C)), LayerNormalization4D.{gamma,beta}   normalizations.py:8-37
  DualPathRNN.{norm,rnn.rnn_lst.N,linear}                             rnn_layers.py:96-129
  MultiHeadSelfAttention2D.{Queries,Keys,Values}.N, attn_concat_proj  attention.py:100-147
  InjectionMultiSum.{local_embedding,global_embedding,global_gate}    layers/fusion.py:25-52
  TDANetBlock.{gateway,projection,downsample_layers,globalatt,fusion_layers,concat_layers,residual_conv}  tdanet.py:34-59
  ATTNFusionCell.{key_embed,value_embed,attention_embed,resize}       layers/fusion.py:210-243
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

EPS = 1e-5


def _holder_forward(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is fused into the HIP kernels of the stage module that owns it (rtfs_net_amd.models.hip_path) and has no forward "
                       "of its own: call encoder / audio_bottleneck / refinement_module / mask_generator / decoder of the AVNet, or observe it with a forward hook "
                       "(served for RTFS blocks and their direct children, the VP block, the CAF cell: rtfs_net_amd.models.stage_views)")


class GlobalLayerNorm(nn.Module):
    def __init__(self, num_channels: int):
        super().__init__()
        self.norm = nn.GroupNorm(1, num_channels, eps=EPS)

    def forward(self, x):
        return self.norm(x)


class LayerNorm4D(nn.Module):
    """gamma/beta of shape [1, C, 1, F] (F == 1: normalise over C only)."""

    def __init__(self, chan: int, freqs: int):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(1, chan, 1, freqs))
        self.beta = nn.Parameter(torch.zeros(1, chan, 1, freqs))

    forward = _holder_forward


def _make_norm(kind, chan):
    if kind is None:
        return nn.Identity()
    if kind == "gLN":
        return GlobalLayerNorm(chan)
    if kind in ("BatchNorm1d", "BatchNorm2d"):
        return getattr(nn, kind)(chan)
    raise ValueError(f"Could not interpret normalization identifier: {kind}")


def _make_act(kind):
    if kind is None:
        return nn.Identity()
    if kind in ("PReLU", "ReLU", "Sigmoid", "Tanh"):
        return getattr(nn, kind)()
    raise ValueError(f"Could not interpret activation identifier: {kind}")


def _conv1d_glue(conv, x):
    """Conv1d of the VP branch without MIOpen: every conv there is pointwise or depthwise (TDANet.py:24-66, 135-160), and at
    [B, <=512, <=Tv] MIOpen falls back to per-sample im2col + GEMM loops (thousands of 4 us launches per training step).
    Pointwise = one batched GEMM; depthwise = pad + strided window view + one multiply-reduce."""
    k, s, w = conv.kernel_size[0], conv.stride[0], conv.weight
    if k == 1 and conv.groups == 1 and s == 1:
        y = torch.matmul(w[:, :, 0], x)
    elif conv.groups == conv.in_channels == conv.out_channels and conv.dilation[0] == 1:
        if k == 1 and s == 1:
            y = x * w.view(1, -1, 1)
        else:
            if isinstance(conv.padding, str):  # 'same', stride 1: total k-1, left (k-1)//2 (torch puts the extra sample on the right)
                lp, rp = (k - 1) // 2, k - 1 - (k - 1) // 2
            else:
                lp = rp = conv.padding[0]
            y = (F.pad(x, (lp, rp)).unfold(-1, k, s) * w.view(1, -1, 1, k)).sum(-1)
    elif k == 1 and s == 1 and conv.groups > 1 and conv.in_channels % conv.groups == 0 and conv.out_channels % conv.groups == 0:
        # grouped pointwise conv (CAF attention_embed / resize: 256 groups of 2 -> 4 / 2 -> 1, layers/fusion.py:208-228): one batched
        # contraction instead of MIOpen's grouped-conv fallbacks (its weight-gradient kernel alone was ~1 ms per training step)
        g_, ci, co = conv.groups, conv.in_channels // conv.groups, conv.out_channels // conv.groups
        y = torch.einsum("bgit,goi->bgot", x.reshape(x.shape[0], g_, ci, x.shape[-1]), w.reshape(g_, co, ci)).reshape(x.shape[0], conv.out_channels, x.shape[-1])
    else:
        return conv(x)
    return y if conv.bias is None else y + conv.bias.view(1, -1, 1)


class ConvNormAct(nn.Module):
    """pre_norm -> pre_act -> conv -> norm -> act, padding 'same' for stride 1 else dil*(k-1)//2."""

    def __init__(self, in_chan, out_chan, kernel_size, *, is2d, stride=1, groups=1, pre_norm_type=None, pre_act_type=None,
                 norm_type=None, act_type=None, bias=True, xavier_init=False):
        super().__init__()
        self.in_chan, self.kernel_size, self.stride, self.groups = in_chan, kernel_size, stride, groups
        self.out_chan = out_chan if kernel_size > 0 else in_chan
        if kernel_size > 0:
            conv_cls = nn.Conv2d if is2d else nn.Conv1d
            pad = (kernel_size - 1) // 2 if stride > 1 else "same"
            conv = conv_cls(in_chan, out_chan, kernel_size, stride=stride, padding=pad, groups=groups, bias=bias)
            if xavier_init:
                nn.init.xavier_uniform_(conv.weight)
        else:
            conv = nn.Identity()
        self.full_layer = nn.Sequential(_make_norm(pre_norm_type, in_chan), _make_act(pre_act_type), conv,
                                        _make_norm(norm_type, self.out_chan), _make_act(act_type))

    def forward(self, x):  # used by the VP (video) branch only
        pre_norm, pre_act, conv, norm, act = self.full_layer
        x = pre_act(pre_norm(x))
        if isinstance(conv, nn.Conv1d):
            x = _conv1d_glue(conv, x)
        else:
            x = conv(x)
        return act(norm(x))

    def get_config(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_") and k != "training"}


class ConvActNorm4D(nn.Module):
    """1x1 Conv2d -> PReLU -> LayerNorm4D over (C, F)."""

    def __init__(self, in_chan, out_chan, n_freqs):
        super().__init__()
        self.conv = nn.Conv2d(in_chan, out_chan, 1)
        self.act = nn.PReLU()
        self.norm = LayerNorm4D(out_chan, n_freqs)

    forward = _holder_forward


class SRUCellParams(nn.Module):
    """weight [d_in, 2*d*k], weight_c [4*d], bias [4*d], scale_x buffer; init as in sru (oracle/sru_ref.py header)."""

    def __init__(self, input_size, hidden_size, highway_bias=0.0, rescale=False):
        super().__init__()
        out = 2 * hidden_size
        k = 3 if input_size == out else 4
        self.weight = nn.Parameter(torch.empty(input_size, out * k))
        self.weight_c = nn.Parameter(torch.empty(2 * out))
        self.bias = nn.Parameter(torch.zeros(2 * out))
        self.register_buffer("scale_x", torch.ones(1))
        with torch.no_grad():
            b = (3.0 / input_size) ** 0.5
            self.weight.uniform_(-b, b)
            w = self.weight.view(input_size, out, k)
            w[:, :, 1].mul_(0.5**0.5)
            w[:, :, 2].mul_(0.5**0.5)
            self.weight_c.uniform_(-(3.0**0.5), 3.0**0.5).mul_(0.5**0.5)
            self.bias[out:].add_(highway_bias)
            if rescale:
                self.scale_x.fill_((1 + math.exp(highway_bias) * 2) ** 0.5)
                if k == 4:
                    w[:, :, 3].mul_(float(self.scale_x))

    forward = _holder_forward


class SRUParams(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, highway_bias=0.0, rescale=False):
        super().__init__()
        self.rnn_lst = nn.ModuleList(
            SRUCellParams(input_size if i == 0 else 2 * hidden_size, hidden_size, highway_bias, rescale) for i in range(num_layers)
        )

    forward = _holder_forward


class DualPathRNN(nn.Module):
    def __init__(self, in_chan, hid_chan, dim, kernel_size=8, stride=1, rnn_type="SRU", num_layers=4, bidirectional=True, **kw):
        super().__init__()
        if rnn_type != "SRU" or not bidirectional or stride != 1:
            raise ValueError("the HIP path supports DualPathRNN with a bidirectional SRU and stride 1")
        self.in_chan, self.hid_chan, self.dim, self.kernel_size, self.num_layers = in_chan, hid_chan, dim, kernel_size, num_layers
        self.norm = LayerNorm4D(in_chan, 1)
        self.rnn = SRUParams(in_chan * kernel_size, hid_chan, num_layers, kw.get("highway_bias", 0.0), kw.get("rescale", False))
        self.linear = nn.ConvTranspose1d(2 * hid_chan, in_chan, kernel_size, stride=stride)

    forward = _holder_forward


class MultiHeadSelfAttention2D(nn.Module):
    def __init__(self, in_chan, n_freqs, n_head=4, hid_chan=4, act_type="PReLU", norm_type="LayerNormalization4D", dim=3, **kw):
        super().__init__()
        assert in_chan % n_head == 0
        if act_type != "PReLU" or norm_type != "LayerNormalization4D" or dim != 3:
            raise ValueError("the HIP path supports MultiHeadSelfAttention2D with PReLU + LayerNormalization4D, dim 3")
        self.in_chan, self.n_freqs, self.n_head, self.hid_chan = in_chan, n_freqs, n_head, hid_chan
        self.Queries = nn.ModuleList(ConvActNorm4D(in_chan, hid_chan, n_freqs) for _ in range(n_head))
        self.Keys = nn.ModuleList(ConvActNorm4D(in_chan, hid_chan, n_freqs) for _ in range(n_head))
        self.Values = nn.ModuleList(ConvActNorm4D(in_chan, in_chan // n_head, n_freqs) for _ in range(n_head))
        self.attn_concat_proj = ConvActNorm4D(in_chan, in_chan, n_freqs)

    forward = _holder_forward


class InjectionMultiSum(nn.Module):
    def __init__(self, in_chan, kernel_size, norm_type, is2d):
        super().__init__()
        kw = dict(is2d=is2d, groups=in_chan, norm_type=norm_type, bias=False)
        self.local_embedding = ConvNormAct(in_chan, in_chan, kernel_size, **kw)
        self.global_embedding = ConvNormAct(in_chan, in_chan, kernel_size, **kw)
        self.global_gate = ConvNormAct(in_chan, in_chan, kernel_size, act_type="Sigmoid", **kw)

    def forward(self, local, glob):  # VP branch (1-D) only
        new, old = local.shape[-1], glob.shape[-1]
        loc = self.local_embedding(local)
        if new > old:
            g = F.interpolate(self.global_embedding(glob), size=new, mode="nearest")
            gate = F.interpolate(self.global_gate(glob), size=new, mode="nearest")
        else:
            gi = F.interpolate(glob, size=new, mode="nearest")
            g, gate = self.global_embedding(gi), self.global_gate(gi)
        return loc * gate + g


class PositionalEncoding(nn.Module):
    def __init__(self, channels, max_len=10000):
        super().__init__()
        pe = torch.zeros(max_len, channels)
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        div = torch.exp(torch.arange(0, channels, 2).float() * -(torch.log(torch.tensor(max_len).float()) / channels))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return x + self.pe[:, : x.size(1)]


class DropPath(nn.Module):
    """Per-sample stochastic depth (what timm.models.layers.DropPath does); identity in eval."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1 - self.p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask.div_(keep)


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, in_chan, n_head=8, dropout=0.1):
        super().__init__()
        assert in_chan % n_head == 0
        self.norm1 = nn.LayerNorm(in_chan)
        self.pos_enc = PositionalEncoding(in_chan)
        self.attention = nn.MultiheadAttention(in_chan, n_head, dropout, batch_first=True)
        self.dropout_layer = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(in_chan)
        self.drop_path_layer = DropPath(dropout)

    def forward(self, x):
        res = x
        x = self.pos_enc(self.norm1(x.transpose(1, 2)))
        x = self.dropout_layer(self.attention(x, x, x, need_weights=False)[0]) + x
        x = self.norm2(x).transpose(2, 1)
        return self.drop_path_layer(x) + res


class FeedForwardNetwork(nn.Module):
    def __init__(self, in_chan, hid_chan, kernel_size, dropout):
        super().__init__()
        self.encoder = ConvNormAct(in_chan, hid_chan, 1, is2d=False, norm_type="gLN", bias=False)
        self.refiner = ConvNormAct(hid_chan, hid_chan, kernel_size, is2d=False, groups=hid_chan, act_type="ReLU")
        self.decoder = ConvNormAct(hid_chan, in_chan, 1, is2d=False, norm_type="gLN", bias=False)
        self.dropout_layer = DropPath(dropout)

    def forward(self, x):
        y = self.dropout_layer(self.refiner(self.encoder(x)))
        return self.dropout_layer(self.decoder(y)) + x


class GlobalAttention(nn.Module):
    def __init__(self, in_chan, hid_chan=None, ffn_name="FeedForwardNetwork", kernel_size=5, n_head=8, dropout=0.1, **kw):
        super().__init__()
        if ffn_name != "FeedForwardNetwork":
            raise ValueError(f"unsupported ffn_name {ffn_name}")
        self.MHSA = MultiHeadSelfAttention(in_chan, n_head, dropout)
        self.FFN = FeedForwardNetwork(in_chan, hid_chan if hid_chan is not None else 2 * in_chan, kernel_size, dropout)

    def forward(self, x):
        return self.FFN(self.MHSA(x))


_LAYER_TYPES = {"DualPathRNN": DualPathRNN, "MultiHeadSelfAttention2D": MultiHeadSelfAttention2D, "GlobalAttention": GlobalAttention}


class TDANetBlock(nn.Module):
    def __init__(self, in_chan, hid_chan, kernel_size, stride, norm_type, act_type, upsampling_depth, layers, is2d):
        super().__init__()
        self.in_chan, self.hid_chan, self.kernel_size, self.stride = in_chan, hid_chan, kernel_size, stride
        self.upsampling_depth, self.is2d = upsampling_depth, is2d
        self.gateway = ConvNormAct(in_chan, in_chan, 1, is2d=is2d, groups=in_chan, act_type=act_type)
        self.projection = ConvNormAct(in_chan, hid_chan, 1, is2d=is2d, norm_type=norm_type, act_type=act_type)
        self.downsample_layers = nn.ModuleList(
            ConvNormAct(hid_chan, hid_chan, kernel_size, is2d=is2d, stride=1 if i == 0 else stride, groups=hid_chan, norm_type=norm_type)
            for i in range(upsampling_depth)
        )
        mods = []
        for _, layer in layers.items():
            cls = _LAYER_TYPES.get(layer["layer_type"])
            if cls is None:
                raise ValueError(f"Could not interpret normalization identifier: {layer['layer_type']}")
            mods.append(cls(in_chan=hid_chan, **{k: v for k, v in layer.items() if k != "layer_type"}))
        self.globalatt = nn.Sequential(*mods)
        self.fusion_layers = nn.ModuleList(InjectionMultiSum(hid_chan, kernel_size, norm_type, is2d) for _ in range(upsampling_depth))
        self.concat_layers = nn.ModuleList(InjectionMultiSum(hid_chan, kernel_size, norm_type, is2d) for _ in range(upsampling_depth - 1))
        self.residual_conv = ConvNormAct(hid_chan, in_chan, 1, is2d=is2d)

    def forward(self, x):
        """Torch execution of the 1-D (video) block: separators/tdanet.py:106-133."""
        if self.is2d:
            _holder_forward(self)
        residual = self.gateway(x)
        ds = [self.downsample_layers[0](self.projection(residual))]
        for i in range(1, self.upsampling_depth):
            ds.append(self.downsample_layers[i](ds[-1]))
        size = ds[-1].shape[-1]
        g = sum(F.adaptive_avg_pool1d(d, size) for d in ds)
        g = self.globalatt(g)
        fused = [self.fusion_layers[i](ds[i], g) for i in range(self.upsampling_depth)]
        exp = self.concat_layers[-1](fused[-2], fused[-1]) + ds[-2]
        for i in range(self.upsampling_depth - 3, -1, -1):
            exp = self.concat_layers[i](fused[i], exp) + ds[i]
        return self.residual_conv(exp) + residual


class TDANet(nn.Module):
    def __init__(self, in_chan=-1, hid_chan=-1, kernel_size=5, stride=2, norm_type="gLN", act_type="PReLU", upsampling_depth=4,
                 layers=None, repeats=4, shared=False, is2d=False, **kw):
        super().__init__()
        self.repeats, self.shared = repeats, shared

        def mk():
            return TDANetBlock(in_chan, hid_chan, kernel_size, stride, norm_type, act_type, upsampling_depth, layers or {}, is2d)

        self.blocks = mk() if shared else nn.ModuleList(mk() for _ in range(repeats))

    def get_block(self, i):
        return self.blocks if self.shared else self.blocks[i]

    forward = _holder_forward


class ATTNFusionCell(nn.Module):
    def __init__(self, in_chan_a, in_chan_b, kernel_size, is2d):
        super().__init__()
        self.in_chan_a, self.in_chan_b, self.kernel_size = in_chan_a, in_chan_b, kernel_size
        self.key_embed = ConvNormAct(in_chan_a, in_chan_a, 1, is2d=is2d, groups=in_chan_a, norm_type="BatchNorm2d", act_type="ReLU", bias=False)
        self.value_embed = ConvNormAct(in_chan_a, in_chan_a, 1, is2d=is2d, groups=in_chan_a, norm_type="BatchNorm2d", bias=False)
        self.attention_embed = ConvNormAct(in_chan_b, kernel_size * in_chan_a, 1, is2d=False, groups=in_chan_a, norm_type="gLN")
        self.resize = ConvNormAct(in_chan_b, in_chan_a, 1, is2d=False, groups=in_chan_a, norm_type="gLN")

    forward = _holder_forward


class ATTNFusion(nn.Module):
    def __init__(self, ain_chan, vin_chan, kernel_size, video_fusion, is2d):
        super().__init__()
        if video_fusion:
            raise ValueError("video-side fusion (video repeats > 1) is not part of the RTFS-Net family")
        self.audio_lstm = ATTNFusionCell(ain_chan, vin_chan, kernel_size, is2d)

    forward = _holder_forward


class MultiModalFusion(nn.Module):
    def __init__(self, audio_bn_chan, video_bn_chan, kernel_size=1, fusion_repeats=3, fusion_type="ConcatFusion", fusion_shared=False,
                 is2d=False, **kw):
        super().__init__()
        if fusion_type != "ATTNFusion":
            raise ValueError(f"Could not interpret fusion identifier: {fusion_type} (the HIP path implements ATTNFusion)")
        if fusion_repeats != 1:
            raise ValueError("the HIP path supports exactly one audio-visual fusion (video_params.repeats == 1)")
        self.fusion_shared = fusion_shared
        cell = ATTNFusion(audio_bn_chan, video_bn_chan, kernel_size, False, is2d)
        self.fusion_module = cell if fusion_shared else nn.ModuleList([cell])

    def get_fusion_block(self, i):
        return self.fusion_module if self.fusion_shared else self.fusion_module[i]

    forward = _holder_forward
