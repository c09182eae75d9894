"""GPU parity of the training step: gradients of EVERY parameter from the HIP backward chain (audio branch, CAF, VP block) against float64
autograd of THE REFERENCE ITSELF (tests/golden/grads_*.npz, written by oracle/gen_golden_grads.py from /root/reference/src/models - which also holds
the oracle restatement's autograd to 1e-7 of the reference's; the GPU box does no CPU-side oracle work for these cases), in eval mode (BatchNorm running statistics) and train mode (batch statistics; dropout forced to 0
because the oracle has none), plus the BatchNorm running statistics the train-mode step leaves behind.

Deterministic: one input seed per case, one tolerance per precision, no retries.
  fp32 / bf16x6 step:  ||g - ref|| <= 3e-3 * (||ref|| + 1e-4 * largest gradient norm) per parameter tensor; the 15 scalar PReLU slopes 3e-3 in
                       eval mode and 1e-2 in train mode (activation kinks, not summation: see _check_parameter_gradients); median over the
                       tensors < 1e-3.  RTFS-Net-6 itself (six passes through the shared block, full length) is the eighth case.
  bf16x3 step:         1.5e-3 per tensor, 3e-3 on the scalar slopes (observed 1.1e-3), median < 1e-3 - on the SMOOTH-REGIME weights of oracle/regimes.py smooth_regime
                       (observed on MI355X: median 1e-5 ... 2e-5, worst tensor 2.8e-4).
Activation kinks: an fp32 evaluation that lands on the other side of a PReLU / ReLU kink than float64 is off by O(1) in that element's
derivative.  One element of an audio tensor is noise, one element of the 50-token video branch moves ~100 tensors by 1e-2 - so every case
runs on lip embeddings that regimes.stable_emb has moved (deterministically, only if needed: the default full-length input has a video
activation 3e-6 of its site's rms away from 0) out of round-off distance of every video-branch kink.  The split-bf16 step carries ~70x the
fp32 round-off and flips thousands of AUDIO activations on ordinary weights; its GEMM entry points are what differs from the fp32 step, and
they are checked where the function is smooth (slopes in [0.97, 1], ReLU inputs positive), tightly, instead of loosely where it is not.
"""
import numpy as np
import pytest
import torch

from oracle.regimes import GRAD_CASES as CASES, GRAD_WEIGHT_SEED, SMOOTH_CASES, case_name
from util import load_npz, make_model, rel, synth

pytestmark = pytest.mark.gpu
GLUE_VIDEO = ("refinement_module.video_net.",)

@pytest.mark.parametrize("training,B,L,R,Tv", CASES)
def test_parameter_gradients(training, B, L, R, Tv):
    _check_parameter_gradients(training, B, L, R, Tv, "f32")


@pytest.mark.parametrize("training,B,L,R,Tv", SMOOTH_CASES)
def test_parameter_gradients_split_bf16_step(training, B, L, R, Tv):
    """`set_compute_dtype("bf16x3")`: forward GEMMs, weight-gradient and input-gradient GEMMs of the adjoint chain as three-term split-bf16
    products on the bf16 MFMA pipe (fp32 accumulation), on the smooth-regime weights (module docstring)."""
    _check_parameter_gradients(training, B, L, R, Tv, "bf16x3")


@pytest.mark.parametrize("training,B,L,R,Tv", [CASES[1], CASES[5]])
def test_parameter_gradients_fp32_equivalent_split_step(training, B, L, R, Tv):
    """`set_compute_dtype("bf16x6")`: every MFMA product of the step as six bf16 products of the three-way split operands - fp32-level
    accuracy, so the fp32 step's tolerance applies unchanged"""
    _check_parameter_gradients(training, B, L, R, Tv, "bf16x6")


def test_parameter_gradients_split_bf16_step_on_ordinary_weights():
    """the split-bf16 step also on ORDINARY weights (active PReLU / ReLU kinks, slopes 0.1 ... 0.4), where its ~70x fp32 round-off flips
    audio activations across their kinks: the documented looser bound (median < 5e-3, per tensor 6e-2) - a wrong operand fed to an activation
    adjoint on the bf16 path shows as O(1) errors, far outside it"""
    _check_parameter_gradients(*CASES[2], "bf16x3", kind="plain", tol=6e-2, tol_scalar=6e-2, tol_median=5e-3)


def test_parameter_gradients_at_the_config3_shape():
    """BASELINE configs[2] at its real shape against the REFERENCE: RTFS-Net-6, batch 32 - every large-batch kernel form of the step runs
    (fast-FIR layer-0 GEMM / ConvTranspose / its input gradient, weight-stationary GEMMs, `rows_ws64`, the >= 2048-sequence SRU forms, the Toeplitz weight
    gradient at 4000 x 57).  Utterance 0 is the input of the reference's RTFS-Net-6 fixture (float64 autograd of /root/reference, oracle/gen_golden_grads.py),
    the other 31 are synthetic; the loss weights are the fixture's for utterance 0 and ZERO for the rest.  Every adjoint is linear in d(out) and nothing
    couples utterances in eval mode (gLN is per utterance, BatchNorm uses running statistics), so all 363 parameter gradients must equal the fixture's."""
    _check_parameter_gradients(*CASES[-1], "f32", embed_in_batch=32)


@pytest.mark.parametrize("training,B,L,R,Tv", SMOOTH_CASES)
def test_parameter_gradients_fp32_step_in_the_smooth_regime(training, B, L, R, Tv):
    """the SHARED-block fp32 step where the function is smooth (slopes in [0.97, 1], ReLU inputs positive: no activation kink within
    round-off of any element), held ten times tighter than on ordinary weights - a 0.5 % operand error in ONE adjoint shows here.  (The ordinary-weight
    cases above keep 3e-3: their deviations are kink flips)"""
    # 3e-4 per tensor and per scalar slope, median 1e-5: ten times inside the ordinary-weight bound.  Observed on MI355X: median 1.6e-6; the worst tensor of the
    # full-length case is a 64 x 4 attention query projection at 1.35e-4 (softmax shift invariance leaves it a cancellation residue), worst slope 1.04e-4
    _check_parameter_gradients(training, B, L, R, Tv, "f32", kind="smooth", tol=3e-4, tol_scalar=3e-4, tol_median=1e-5)


def _check_parameter_gradients(training, B, L, R, Tv, dtype, kind=None, tol=None, tol_scalar=None, tol_median=1e-3, embed_in_batch=None):
    model, sd, cfg = make_model(R, "cuda")
    for mod in model.modules():
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    kind = kind or ("smooth" if dtype == "bf16x3" else "plain")
    z = load_npz(case_name(kind, training, B, L, R, Tv) + ".npz")
    mix, _, _ = synth.synth_inputs(B, L, Tv)
    assert np.array_equal(mix[:, :256].numpy(), z["mix_head"])
    if kind == "smooth":  # smooth-regime weights (oracle/regimes.py): the fixture holds the entries that differ from the synthetic state dict
        sd = dict(sd)
        sd.update({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
        model.load_state_dict(sd)
    emb = torch.from_numpy(z["emb"])  # kink-stable lip embeddings (regimes.stable_emb)
    model.train(training)
    model.set_compute_dtype(dtype)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(GRAD_WEIGHT_SEED))
    if embed_in_batch:  # the fixture's utterances in front of synthetic ones that carry zero loss weight
        assert not training and embed_in_batch > B
        mix_all, _, emb_all = synth.synth_inputs(embed_in_batch, L, Tv)
        mix_all[:B], emb_all[:B] = mix, emb
        mix, emb, wgt = mix_all, emb_all, torch.cat([wgt, torch.zeros(embed_in_batch - B, 1, L)])
    out = model(mix.cuda(), emb.cuda())
    (out * wgt.cuda()).sum().backward()
    out = out[:B]
    ref_out = torch.from_numpy(z["out"])
    ref = {k[5:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("grad.")}
    ref_stats = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("stat.")}
    assert rel(out.detach(), ref_out) < 1e-3
    tol_video = None
    if tol is None:
        # fp32 / bf16x6: 3e-3 per tensor.  Scalar PReLU slopes: 3e-3 in eval mode (observed <= 1.4e-3 over the five eval cases), 1e-2 in train mode (observed
        # 4.8e-3).  Round 5 measured what these deviations ARE (tools/slope_grad_probe.py, tools/grad_video_bisect.py): not summation error - float64
        # accumulators for the slope sums changed no digit - but activation kinks.  The oracle's OWN fp32 autograd (torch CPU) against the same float64
        # fixtures sits at 1e-6 on the well-conditioned cases exactly like the HIP step (B1_L12100_R1: 1.1e-6, B1_L4096_R3: 2.2e-6 median) and at
        # 1e-3 (tensors) / 2.3e-3 (slopes) on the small train-mode cases, where batch statistics over 2 x (1..16) positions amplify single flips;
        # on the full-length utterance 89 % of the whole deviation of d(block 0 output) is ONE element (the CAF key ReLU).
        tol, tol_scalar = (1.5e-3, 3e-3) if dtype == "bf16x3" else (3e-3, 1e-2 if training else 3e-3)  # (bf16x3 on the smooth-regime weights: observed worst 2.8e-4)
        if dtype == "bf16x6":
            # the six-term step rounds differently from the fp32 step (both to fp32 accuracy), so it flips a DIFFERENT set of kinks on these plain weights: when
            # its layer-0 GEMM changed kernels in round 5, one slope (globalatt.2.Values.1.act of B1_L4096_R3, eval) moved from 1.4e-3 to 4.4e-3 - the size the
            # fp32 step shows in train mode.  The eval-mode 3e-3 above is an observation about ONE rounding, not a property of the arithmetic.
            tol_scalar = 1e-2
        if R >= 6:
            # RTFS-Net-6: every video-branch gradient descends from (d att, d rsz), 25.6 k numbers that each sum 645 elements of the audio gradient at
            # the CAF cell - flipped audio kinks of the five blocks behind it land there undiluted and spread over ALL video tensors alike
            # (observed 3.4e-3 on the worst one, 1.6e-4 median over the model; the oracle's own fp32 autograd: 1.7e-4 / 7.4e-5)
            tol_video = 6e-3
    glue_video = training and Tv < 8  # the VP block as PyTorch glue on <= 7 tokens with batch statistics over B x (1 ... 6) positions: not a kernel of this build
    scale = max(float(g.norm()) for g in ref.values())
    checked, errs, bad, errs_scalar, errs_tensor = 0, [], [], [], []
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        if float(ref[n].norm()) < 1e-6 * scale:
            assert float(p.grad.norm()) < 1e-4 * scale, n  # analytically zero gradients (softmax shift invariance, bias before BatchNorm)
            continue
        if glue_video and n.startswith(GLUE_VIDEO):
            continue
        # mixed tolerance (as allclose): tensors whose whole gradient is ~1e-4 of the largest one are cancellation residue of fp32 sums
        # (softmax over Tv, BatchNorm) and are held to the absolute floor instead
        err = float((p.grad.double().cpu() - ref[n]).norm()) / (float(ref[n].norm()) + 1e-4 * scale)
        if err >= (tol_scalar if p.numel() <= 12 else (tol_video if (tol_video and n.startswith(GLUE_VIDEO)) else tol)):
            bad.append((round(err, 5), n))
        errs.append(err)
        (errs_scalar if p.numel() <= 12 else errs_tensor).append(err)
        checked += 1
    assert checked > 150
    errs.sort()
    print(f"{dtype} train={training} B={B} L={L} R={R} Tv={Tv}: median gradient error {errs[len(errs) // 2]:.2e}, worst tensor {max(errs_tensor):.2e}, "
          f"worst scalar slope {max(errs_scalar):.2e}, {checked} tensors")
    assert not bad, sorted(bad, reverse=True)[:12]
    assert errs[len(errs) // 2] < tol_median
    if training:  # running statistics of the 26 VP BatchNorm1d + 2 CAF BatchNorm2d layers after one step (momentum 0.1, unbiased variance)
        got = {k: v for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
        assert len(got) == 56 and set(got) == set(ref_stats)
        for k, v in got.items():
            if glue_video and k.startswith(GLUE_VIDEO) and k.endswith("running_var"):
                continue  # (unbiased variance over 2 positions: n / (n - 1) of a difference of two fp32 numbers)
            assert rel(v, ref_stats[k]) < 1e-4, (k, rel(v, ref_stats[k]))


def test_nonshared_blocks_forward_and_parameter_gradients():
    """audio_params.shared = False (tdanet.py:170-181,201-205: R RTFS blocks with their own weights; no shipped config uses it): waveform and every parameter
    gradient of BOTH blocks against the reference's float64 autograd (tests/golden/grads_nonshared_*.npz, written by oracle/gen_golden_grads.py from the
    reference with that flag; the oracle restatement agrees with it to 1e-15)."""
    import copy

    from oracle.regimes import NONSHARED_CASES
    from rtfs_net_amd import AVNet

    training, B, L, R, Tv = NONSHARED_CASES[0]
    cfg = synth.rtfs_audionet(R)
    cfg["audio_params"]["shared"] = False
    model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
    sd = synth.synth_state_dict(model.state_dict())
    assert "refinement_module.audio_net.blocks.1.gateway.full_layer.2.weight" in sd
    z = load_npz(case_name("nonshared", training, B, L, R, Tv) + ".npz")
    sd.update({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})  # smooth-regime weights (oracle/regimes.py NONSHARED_CASES says why)
    model.load_state_dict(sd)
    model = model.cuda()
    mix, _, _ = synth.synth_inputs(B, L, Tv)
    emb = torch.from_numpy(z["emb"])
    with torch.no_grad():
        assert rel(model(mix.cuda(), emb.cuda()), torch.from_numpy(z["out"])) < 1e-5  # inference path (per-block weights, no projection fusion)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(GRAD_WEIGHT_SEED))
    out = model(mix.cuda(), emb.cuda())
    assert rel(out.detach(), torch.from_numpy(z["out"])) < 1e-5
    (out * wgt.cuda()).sum().backward()
    ref = {k[5:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("grad.")}
    scale = max(float(g.norm()) for g in ref.values())
    errs, per_block = [], {0: 0, 1: 0}
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        if float(ref[n].norm()) < 1e-6 * scale:
            continue
        err = float((p.grad.double().cpu() - ref[n]).norm()) / (float(ref[n].norm()) + 1e-4 * scale)
        assert err < (1e-3 if p.numel() <= 12 else 3e-4), (n, err)  # (smooth regime, fp32 step: observed median 1.3e-6, worst 1.7e-5)
        errs.append(err)
        for bi in (0, 1):
            per_block[bi] += n.startswith(f"refinement_module.audio_net.blocks.{bi}.")
    errs.sort()
    assert per_block[0] > 100 and per_block[0] == per_block[1] and errs[len(errs) // 2] < 1e-3
    print(f"non-shared blocks: {len(errs)} tensors, median {errs[len(errs) // 2]:.2e}, worst {errs[-1]:.2e}")


def test_input_of_caf_video_side_gets_gradient():
    """the Function returns d(att), d(rsz): the lip-embedding input must receive a gradient through the torch glue"""
    model, _, _ = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(1, 4096, 6)
    e = emb.cuda().requires_grad_(True)
    model(mix.cuda(), e).square().mean().backward()
    assert e.grad is not None and torch.isfinite(e.grad).all() and float(e.grad.abs().max()) > 0


def test_video_branch_backward_after_the_caller_dropped_the_embedding():
    """Round 6 (found by looping the suite's first test in fresh processes: 27 wrong runs of 60 on a warm box, d(video gateway weight) off by 100 %): the video
    branch runs on a side stream, forward and backward, and its backward reads the CALLER's lip-embedding tensor (saved by autograd: the gateway's weight gradient
    is d(out) x that tensor).  A caller that drops the tensor - `model(mix, emb.cuda())` - hands its block back to the main stream's allocator pool as soon as the
    last side-stream node is enqueued; the running backward's next main-stream allocation can then overwrite it before the side-stream kernel has read it.
    `AVNet._forward_autograd` now records the side stream on the tensor.  Reproduced deterministically: the side stream is kept busy (`torch.cuda._sleep`) while the
    backward is enqueued, and the main stream churns through small allocations - without the record, the gradient of the video gateway is garbage."""
    training, B, L, R, Tv = CASES[0]
    model, sd, cfg = make_model(R, "cuda")
    z = load_npz(case_name("plain", training, B, L, R, Tv) + ".npz")
    mix, _, _ = synth.synth_inputs(B, L, Tv)
    model.train(training)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(GRAD_WEIGHT_SEED)).cuda()
    mix = mix.cuda()
    model(mix, torch.from_numpy(z["emb"]).cuda()).sum().backward()  # (creates the side stream)
    model.zero_grad(set_to_none=True)
    out = model(mix, torch.from_numpy(z["emb"]).cuda())  # the embedding tensor is a temporary: only autograd's saved tensors keep it alive
    loss = (out * wgt).sum()
    with torch.cuda.stream(model._glue_stream):
        torch.cuda._sleep(200_000_000)  # ~0.1 s: the backward's side-stream nodes queue behind it while the host enqueues the rest
    loss.backward()
    junk = [torch.full((B * 512 * Tv,), 7.0e3, device="cuda") for _ in range(256)]  # main-stream allocations of the embedding's size class while the side stream sleeps
    torch.cuda.synchronize()
    del junk
    name = "refinement_module.video_net.blocks.gateway.full_layer.2.weight"
    got, ref = dict(model.named_parameters())[name].grad, torch.from_numpy(z["grad." + name])
    assert rel(got, ref) < 3e-3, rel(got, ref)
