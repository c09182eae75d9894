"""CPU experiment behind the bf16 design: the oracle with every dense contraction's operands rounded to bfloat16
(fp32 accumulation), in two flavours, against the fp32 oracle on the same inputs:

  bf16    one product per operand pair:      a_hi * b_hi
  bf16x3  three-term split:                  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi    (a_lo = bf16(a - a_hi))

Contractions touched = the ones the HIP path runs on MFMA: 1x1 convolutions (bottleneck, projection, attention Q/K/V/out, residual
conv, S3 mask), the SRU input GEMMs, ConvTranspose1d, QK^T and PV, the decoder's transposed conv.  Depth-wise convs, norms,
the recurrence and the (i)STFT stay fp32.

    python tools/bf16_error_model.py [layers] [seconds]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import make_model, rel, synth  # noqa: E402

MODE = {"terms": 0}
_conv2d, _conv1d, _convt1d, _convt2d, _matmul = F.conv2d, F.conv1d, F.conv_transpose1d, F.conv_transpose2d, torch.Tensor.__matmul__


def split(x):
    hi = x.bfloat16().float()
    return hi, (x - hi).bfloat16().float()


def contract(fn, a, b):
    if MODE["terms"] == 0:
        return fn(a, b)
    ah, al = split(a)
    bh, bl = split(b)
    if MODE["terms"] == 1:
        return fn(ah, bh)
    return fn(ah, bh) + fn(ah, bl) + fn(al, bh)


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if groups == 1 and w.shape[-1] == 1 and w.shape[-2] == 1:
        y = contract(lambda p, q: _conv2d(p, q), x, w)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return _conv2d(x, w, b, stride, padding, dilation, groups)


def conv1d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    return _conv1d(x, w, b, stride, padding, dilation, groups)  # video branch: fp32 (one workgroup per utterance, VALU)


def convt1d(x, w, b=None, stride=1, **kw):
    y = contract(lambda p, q: _convt1d(p, q, None, stride), x, w)
    return y if b is None else y + b.view(1, -1, 1)


def convt2d(x, w, b=None, stride=1, padding=0, **kw):
    return contract(lambda p, q: _convt2d(p, q, None, stride, padding), x, w)


def matmul(a, b):
    if a.is_floating_point() and b.is_floating_point() and a.dim() >= 2 and b.dim() >= 2:
        return contract(_matmul, a, b)
    return _matmul(a, b)


def main(R=12, seconds=4.0):
    from oracle import avnet_ref, sru_ref

    torch.set_num_threads(8)
    L = int(16000 * seconds)
    _, sd, cfg = make_model(R)
    mix, _, emb = synth.synth_inputs(1, L, int(25 * seconds))
    with torch.no_grad():
        ref = avnet_ref.avnet_forward(sd, cfg, mix, emb)
    F.conv2d, F.conv1d, F.conv_transpose1d, F.conv_transpose2d, torch.Tensor.__matmul__ = conv2d, conv1d, convt1d, convt2d, matmul
    try:
        for terms, name in ((1, "bf16"), (3, "bf16x3")):
            MODE["terms"] = terms
            with torch.no_grad():
                out = avnet_ref.avnet_forward(sd, cfg, mix, emb)
            print(f"RTFS-Net-{R}, {seconds:g} s: {name:7s} waveform rel L2 vs the fp32 oracle = {rel(out, ref):.3e}")
    finally:
        F.conv2d, F.conv1d, F.conv_transpose1d, F.conv_transpose2d, torch.Tensor.__matmul__ = _conv2d, _conv1d, _convt1d, _convt2d, _matmul


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12, float(sys.argv[2]) if len(sys.argv) > 2 else 4.0)
