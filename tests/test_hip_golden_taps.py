"""GPU: the HIP stage boundaries of EVERY block of a full-size forward against the REFERENCE's own values.

tests/golden/rtfs6_b2.npz (oracle/gen_golden.py, produced by running /root/reference) holds, for RTFS-Net-6 on B = 2 utterances of
2 s, a 4096-point strided sample and the norm of each forward-hook output of each of the 6 block applications.  The HIP path
captures the same boundaries (`HipForward.tap_all_blocks`); blocks 2..R-1 take their projection from the fused
residual + projection kernel of the previous block, so this is also that kernel's check against the reference.

Tolerance: relative L2 of the strided sample <= 5e-4 (fp32 chain error through up to 6 blocks; observed ~1e-6), norm within 5e-4.
"""
import pytest
import torch
import torch.nn.functional as F

from util import cl_to_nchw, load_npz, make_model, rel, synth

pytestmark = pytest.mark.gpu
TOL = 5e-4


def strided(t, n=4096):
    f = t.flatten()
    return f[:: max(1, f.numel() // n)][:n]


@pytest.mark.parametrize("name,R,B,L", [("rtfs6_b2.npz", 6, 2, 32000), ("rtfs12_4s_b1.npz", 12, 1, 64000)])
def test_every_block_stage_against_reference_taps(name, R, B, L):
    z = load_npz(name)
    model, sd, _ = make_model(R, "cuda")
    Tv = 25 * L // 16000
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    model._hip.taps, model._hip.tap_all_blocks = {}, True
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
    torch.cuda.synchronize()
    taps = model._hip.taps
    model._hip.taps, model._hip.tap_all_blocks = None, False
    assert rel(out, torch.from_numpy(z["out"])) < 1e-3
    T = 1 + L // 128
    T2 = (T - 2) // 2 + 1
    full = lambda t, C: cl_to_nchw(t.float().cpu(), B, T, 129, C)  # noqa: E731
    low = lambda t: cl_to_nchw(t.float().cpu(), B, T2, 64, 64)  # noqa: E731
    p = "refinement_module.audio_net.blocks."
    gln = lambda x, q: F.group_norm(x, 1, sd[q + "3.norm.weight"], sd[q + "3.norm.bias"], 1e-5)  # noqa: E731

    got = {"a_emb": full(taps["a_emb"], 256), "a0": full(taps["a0"], 256), "masked": full(taps["masked"], 256),
           "vp": taps["vp"].float().cpu(), "block": full(taps["block0"], 256)}
    got["caf"] = full(taps["caf_plus_a0"], 256) - got["a0"]
    for i in range(R):
        s = "" if i == 0 else f"#{i}"
        if i > 0:  # the residual kernel of blocks 1..R-2 writes the NEXT block's input, block output + a0 (refinement_module.py:60)
            got["block" + s] = full(taps["block" + s], 256) - (got["a0"] if i < R - 1 else 0)
        got["projection" + s] = F.prelu(gln(full(taps["y0" + s], 64), p + "projection.full_layer."), sd[p + "projection.full_layer.4.weight"])
        got["down0" + s] = gln(full(taps["D0" + s], 64), p + "downsample_layers.0.full_layer.")
        got["down1" + s] = gln(low(taps["D1" + s]), p + "downsample_layers.1.full_layer.")
        got["dp_freq" + s], got["dp_time" + s], got["attn" + s] = low(taps["dp_freq" + s]), low(taps["dp_time" + s]), low(taps["attn" + s])
        got["tfar0" + s], got["tfar1" + s] = full(taps["tfar0" + s], 64), low(taps["tfar1" + s])
    assert len(got) == 5 + 9 * R
    worst = ("", 0.0)
    for k, v in got.items():
        e = rel(strided(v), torch.from_numpy(z["tap." + k]))
        n = abs(float(v.double().norm()) / float(z["norm." + k]) - 1)
        worst = max(worst, (k, max(e, n)), key=lambda kv: kv[1])
        assert e < TOL and n < TOL, (k, e, n)
    print("worst stage:", worst)
