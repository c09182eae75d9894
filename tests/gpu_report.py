"""Print the relative L2 error of every HIP stage boundary against the oracle (debug aid; run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
from util import cl_to_nchw, make_model, rel, synth


def main(B=2, L=16000, R=2):
    from oracle.avnet_ref import avnet_forward, stft_frames

    Tv = max(2, 25 * L // 16000)
    model, sd, cfg = make_model(R, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    model._hip.taps = {}
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
    torch.cuda.synchronize()
    t = {k: v.detach().float().cpu() for k, v in model._hip.taps.items()}
    o = {}
    with torch.no_grad():
        ref = avnet_forward(sd, cfg, mix, emb, taps=o)
    T = 1 + L // 128
    T2 = (T - 2) // 2 + 1
    full = lambda n, C: cl_to_nchw(t[n], B, T, 129, C)
    low = lambda n: cl_to_nchw(t[n], B, T2, 64, 64)
    p = "refinement_module.audio_net.blocks."
    rows = [
        ("stft", full("spec", 2), stft_frames(mix, 256, 128)),
        ("a_emb", full("a_emb", 256), o["a_emb"]),
        ("a0", full("a0", 256), o["a0"]),
        ("proj", F.prelu(F.group_norm(full("y0", 64), 1, sd[p + "projection.full_layer.3.norm.weight"], sd[p + "projection.full_layer.3.norm.bias"], 1e-5),
                         sd[p + "projection.full_layer.4.weight"]), o["block0.proj"]),
        ("ds0", F.group_norm(full("D0", 64), 1, sd[p + "downsample_layers.0.full_layer.3.norm.weight"], sd[p + "downsample_layers.0.full_layer.3.norm.bias"], 1e-5), o["block0.ds0"]),
        ("ds1", F.group_norm(low("D1"), 1, sd[p + "downsample_layers.1.full_layer.3.norm.weight"], sd[p + "downsample_layers.1.full_layer.3.norm.bias"], 1e-5), o["block0.ds1"]),
        ("pooled", low("pooled"), o["block0.pooled"]),
        ("dp_freq", low("dp_freq"), o["block0.globalatt.0"]),
        ("dp_time", low("dp_time"), o["block0.globalatt.1"]),
        ("attn", low("attn"), o["block0.globalatt.2"]),
        ("tfar0", full("tfar0", 64), o["block0.fused0"]),
        ("tfar1", low("tfar1"), o["block0.fused1"]),
        ("block0", full("block0", 256), o["block0"]),
        ("vp", t["vp"], o["vp"]),
        ("caf", full("caf_plus_a0", 256) - full("a0", 256), o["caf"]),
        ("refined", full("refined", 256), o[f"block{R-1}"]),
        ("masked", full("masked", 256), o["masked"][:, 0]),
        ("waveform", out.cpu(), ref),
    ]
    for name, a, b in rows:
        print(f"{name:10s} rel={rel(a, b):.3e}  |ref|={float(b.norm()):.4g} nan={bool(torch.isnan(a).any())}", flush=True)


if __name__ == "__main__":
    main()
