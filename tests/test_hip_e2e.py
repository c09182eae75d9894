"""GPU parity of the whole separation path through the module API.

* against the REFERENCE's own outputs (tests/golden/rtfs*.npz, produced by oracle/gen_golden.py from /root/reference)
  -- RTFS-Net-4 (B=1), RTFS-Net-6 (B=2), RTFS-Net-12 on 4 s (B=1): relative L2 <= 1e-3 on the waveform (BASELINE.json)
* size-independent properties at the bench size (B=16): batch invariance, finite output, SI-SDR parity vs the oracle
  on the utterances the oracle can afford (<= 0.01 dB)
* edge cases of the boundary: 1-D / 3-D inputs, odd lengths, ragged Tv, too-short input
"""
import numpy as np
import pytest
import torch

from util import load_npz, make_model, rel, synth

pytestmark = pytest.mark.gpu
WAVE_TOL = 1e-3


@pytest.mark.parametrize("name,R,B,L", [("rtfs4_b1.npz", 4, 1, 32000), ("rtfs6_b2.npz", 6, 2, 32000), ("rtfs12_4s_b1.npz", 12, 1, 64000)])
def test_against_reference_golden(name, R, B, L):
    z = load_npz(name)
    model, _, _ = make_model(R, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, 25 * L // 16000)
    assert np.array_equal(mix[:, :256].numpy(), z["mix_head"])
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
    assert rel(out, torch.from_numpy(z["out"])) < WAVE_TOL


def test_bench_shape_against_reference_golden():
    """BASELINE configs[2] shape (RTFS-Net-6, batch 32, 2 s): at this batch the layer-0 GEMM and the ConvTranspose1d run on the fast-FIR weight-stationary
    kernels (csrc/dualpath.hip unfold_ffa_kernel, round 5) and every 256-wide GEMM on its weight-stationary form.  The first two utterances of the batch are
    the inputs of the reference's own fixture (tests/golden/rtfs6_b2.npz, written by oracle/gen_golden.py from /root/reference): their rows must be the
    reference's waveforms, whatever the other 30 utterances are (utterances are independent)."""
    z = load_npz("rtfs6_b2.npz")
    model, _, _ = make_model(6, "cuda")
    mix2, _, emb2 = synth.synth_inputs(2, 32000, 50)
    assert np.array_equal(mix2[:, :256].numpy(), z["mix_head"])
    mix30, _, emb30 = synth.synth_inputs(30, 32000, 50, seed=synth.INPUT_SEED + 17)
    mix, emb = torch.cat([mix2, mix30], 0), torch.cat([emb2, emb30], 0)
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
        direct = None
        model._hip.variants["unfold"] = 3  # the direct weight-stationary layer-0 kernel of round 4 for comparison
        try:
            direct = model(mix.cuda(), emb.cuda())
        finally:
            model._hip.variants["unfold"] = 0
    ref = torch.from_numpy(z["out"])
    e_ffa, e_dir = rel(out[:2], ref), rel(direct[:2], ref)
    print(f"bench shape vs the reference's waveform: fast-FIR kernels {e_ffa:.2e}, direct layer-0 kernel {e_dir:.2e}")
    assert e_ffa < 2e-5 and e_dir < 2e-5  # (bound 1e-3; the batch-2 forward of the same utterances sits at ~1e-6)
    assert torch.isfinite(out).all() and rel(out, direct) < 1e-5


def test_batch_invariance_and_sisdr_parity():
    from oracle.avnet_ref import avnet_forward, si_sdr

    B, L, Tv = 16, 32000, 50
    model, sd, cfg = make_model(6, "cuda")
    mix, s1, emb = synth.synth_inputs(B, L, Tv)
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
        assert torch.isfinite(out).all()
        # utterances are independent: item 5 alone gives the same waveform as inside the batch
        solo = model(mix[5:6].cuda(), emb[5:6].cuda())
        assert rel(solo[0], out[5]) < 1e-5
        ref = avnet_forward(sd, cfg, mix[5:6], emb[5:6])
    assert rel(out[5:6], ref) < WAVE_TOL
    d = si_sdr(out[5, 0].cpu(), s1[5]) - si_sdr(ref[0, 0], s1[5])
    assert abs(float(d)) < 0.01  # dB


def test_input_shapes_and_edges():
    model, _, _ = make_model(4, "cuda")
    mix, _, emb = synth.synth_inputs(2, 8000 + 37, 11)  # L not a multiple of the hop, ragged Tv
    with torch.no_grad():
        a = model(mix.cuda(), emb.cuda())
        b = model(mix.cuda().unsqueeze(1), emb.cuda())        # [B,1,L]
        c = model(mix[0].cuda(), emb[:1].cuda())               # [L]
        assert a.shape == (2, 1, 8037) and torch.equal(a, b) and rel(c[0], a[0]) < 1e-5
        with pytest.raises(ValueError):
            model(mix[:, :1000].cuda(), emb.cuda())


def test_edge_lengths_against_oracle():
    from oracle.avnet_ref import avnet_forward

    model, sd, cfg = make_model(4, "cuda")
    for L, Tv in ((1920 + 128 * 1, 3), (8037, 11)):
        mix, _, emb = synth.synth_inputs(1, L, Tv)
        with torch.no_grad():
            out = model(mix.cuda(), emb.cuda())
            ref = avnet_forward(sd, cfg, mix, emb)
        assert rel(out, ref) < WAVE_TOL, (L, Tv)


def test_train_mode_runs_with_and_without_autograd():
    """train() + grad enabled = the HIP training step (tests/test_hip_backward.py checks its numbers); train() under no_grad (BatchNorm batch
    statistics + running-statistics update + dropout, nothing to differentiate - what the reference's modules do in that state) runs the same
    forward and drops the graph: same waveform (dropout off so both calls see one function), same running statistics, no grad_fn."""
    model, sd, _ = make_model(4, "cuda")
    for mod in model.modules():
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    model.train()
    mix, _, emb = synth.synth_inputs(2, 8000, 12)
    out = model(mix.cuda(), emb.cuda())
    assert out.requires_grad and out.grad_fn is not None and torch.isfinite(out).all()
    stats = {k: v.clone() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    model.load_state_dict(sd)
    model.train()
    with torch.no_grad():
        out2 = model(mix.cuda(), emb.cuda())
    assert not out2.requires_grad and out2.grad_fn is None and torch.equal(out2, out.detach())
    for k, v in model.state_dict().items():
        if k in stats:
            assert torch.equal(v, stats[k]), k
    assert any(not torch.equal(stats[k], sd[k].to(stats[k].device)) for k in stats if k.endswith("running_mean"))  # (the step did move them)
    # a 10-s segment (625 compressed frames: the attention adjoint walks its keys in two blocks; round 2 refused > 8 s; Tv = 250: VP glue path)
    mix, _, emb = synth.synth_inputs(1, 160000, 250)
    model.zero_grad(set_to_none=True)
    model(mix.cuda(), emb.cuda()).square().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())


def _against_long_fixture(model, name, L, Tv, dtype="f32"):
    """waveform of one long utterance against the REFERENCE's own (tests/golden/long_*.npz, oracle/gen_golden_long.py: every 8th sample + the
    full-length norm), so the GPU box spends no host time on an oracle forward of 8 ... 120 s"""
    z = load_npz(name + ".npz")
    mix, _, emb = synth.synth_inputs(1, L, Tv)
    assert np.array_equal(mix[:, :256].numpy(), z["mix_head"])
    model.set_compute_dtype(dtype)
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
    model.set_compute_dtype("f32")
    assert out.shape == (1, 1, L) and bool(torch.isfinite(out).all())
    e = rel(out[0, 0, ::int(z["stride"])], torch.from_numpy(z["out_strided"]))
    e_norm = abs(float(out.double().norm()) / float(z["norm"]) - 1)
    print(f"{name}, {dtype}: waveform rel L2 vs the reference = {e:.3e} (norm off by {e_norm:.1e})")
    assert e < WAVE_TOL and e_norm < WAVE_TOL
    return mix, emb


def test_long_utterance_8s():
    """8.5 s (531 compressed frames: the 1024-key single-tile attention kernel) against the reference"""
    model, _, _ = make_model(2, "cuda")
    _against_long_fixture(model, "long_8s_R2", 136000, 212)


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_30s_utterance_key_blocked_attention(dtype):
    """30 s = 1875 compressed frames: past the 1024-key score tile the attention core walks the keys in blocks (two-sweep online
    softmax, csrc/attention.hip attn_core_long_kernel); the reference has no length limit.  Against the reference, 1e-3 on the waveform."""
    model, _, _ = make_model(2, "cuda")
    _against_long_fixture(model, "long_30s_R2", 480000, 750, dtype)


def test_64s_utterance():
    """64 s (T = 8001: 1.057e9 bytes per [T][129][256] activation; 4000 compressed frames = four key blocks; Tv = 1600: the CAF video kernel far
    past the 750 frames of the 30 s case, the VP block on the multi-launch chain) against the reference.  One block (the offsets a block
    computes do not depend on the block index)."""
    model, _, _ = make_model(1, "cuda")
    _against_long_fixture(model, "long_64s_R1", 1024000, 1600)


def test_120s_utterance_at_the_length_guard():
    """120 s (T = 15001: 1.98e9 bytes per [T][129][256] activation, just inside the 2 GiB the in-utterance 32-bit byte offsets can address - the
    length guard of models/hip_path.py; 7500 compressed frames = eight key blocks; Tv = 3000) against the reference; 131.1 s (T = 16385) is refused,
    not mis-addressed."""
    model, _, _ = make_model(1, "cuda")
    mix, emb = _against_long_fixture(model, "long_120s_R1", 1920000, 3000)
    with torch.no_grad(), pytest.raises(ValueError):
        model(torch.cat([mix, mix[:, :177152]], 1).cuda(), emb.cuda())
    with torch.no_grad(), pytest.raises(ValueError):  # the guard is the tested envelope itself: one frame more (T = 15002) is refused as well
        model(torch.cat([mix, mix[:, :128]], 1).cuda(), emb.cuda())


def test_batch_offsets_past_2_31_elements():
    """18 utterances of 30 s: B x T x 129 x 256 = 2.23e9 elements > 2^31, so every kernel's batch offset must be 64-bit.  Size-independent
    property: utterances are independent, so the LAST utterance of the batch equals its own batch-1 forward (checked against the oracle by
    test_30s_utterance_key_blocked_attention)."""
    model, _, _ = make_model(1, "cuda")
    B, L, Tv = 18, 480000, 750
    assert B * (1 + L // 128) * 129 * 256 > 2 ** 31
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())[-1].clone()
        torch.cuda.empty_cache()
        one = model(mix[-1:].cuda(), emb[-1:].cuda())[0]
    assert torch.isfinite(out).all()
    assert rel(out, one) < 2e-6


def test_fused_tfar_mix_convolution_matches_the_unfused_pair():
    """rtfs_dwconv_mix_fwd (TFAR mix formed inside the concat-layer convolution's staging) against rtfs_tfar_mix_fwd + rtfs_dwconv_fwd"""
    model, _, _ = make_model(3, "cuda")
    mix, _, emb = synth.synth_inputs(2, 16000 + 77, 25)
    with torch.no_grad():
        fused = model(mix.cuda(), emb.cuda())
        model._hip.fuse["mix"] = False
        plain = model(mix.cuda(), emb.cuda())
    assert rel(fused, plain) < 1e-6
