"""GPU: the BASELINE.json configurations at their FULL sizes, through size-independent properties (the oracle cannot afford them).

* config 2 (RTFS-Net-4, batch 16, forward): batch invariance - an utterance separated alone equals its row of the batch.
* config 3 (RTFS-Net-6, batch 32, forward + backward): the parameter gradients of one B = 32 step equal the SUM of the gradients of
  the two B = 16 half-batch steps (the loss is a sum over utterances and, in eval mode, nothing couples utterances), per tensor
  <= 3e-3 of its norm (1e-2 for the 15 scalar slopes) - the B = 32 launch takes different kernels / tile schedules than any small gradient test (flattened 2- and
  3-sequence tiles, row-split Toeplitz weight gradient, spread accumulators at 8096 workgroups).  One utterance's gradients at the
  full length are checked against float64 autograd of the oracle in tests/test_hip_backward.py (fifth case).
* the SRU skip scaling `scale_x != 1` (a persistent buffer of the reference's state dict) through the whole path:
  PreparedWeights -> rtfs_sru_scan_fwd / rtfs_sru_layer_fwd, against the oracle.
"""
import warnings

import pytest
import torch

from util import make_model, rel, synth

pytestmark = pytest.mark.gpu
SCALAR_TOL = 1e-2  # scalar PReLU slopes (round 3: error x 0.25 against 5e-3; round 4: observed 4.4e-3 here, worst tensor 1.0e-3)


def test_config2_rtfs4_batch16_forward_batch_invariance():
    B, L, Tv = 16, 32000, 50
    model, _, _ = make_model(4, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
        assert out.shape == (B, 1, L) and torch.isfinite(out).all()
        for j in (0, 7, 15):
            solo = model(mix[j:j + 1].cuda(), emb[j:j + 1].cuda())
            assert rel(solo[0], out[j]) < 1e-5, j


def _grads(model, mix, emb, wgt):
    model.zero_grad(set_to_none=True)
    out = model(mix, emb)
    (out * wgt).sum().backward()
    torch.cuda.synchronize()
    return out.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def test_config3_rtfs6_batch32_training_step_is_sum_of_half_batches():
    B, L, Tv, R = 32, 32000, 50, 6
    model, _, _ = make_model(R, "cuda")
    model.eval()  # running-statistics BatchNorm, no dropout: utterances are independent, gradients add
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(3))
    mix, emb, wgt = mix.cuda(), emb.cuda(), wgt.cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "eval() with autograd enabled takes the training-step path": exactly what is wanted here
        out, g32 = _grads(model, mix, emb, wgt)
        o_a, g_a = _grads(model, mix[:16], emb[:16], wgt[:16])
        o_b, g_b = _grads(model, mix[16:], emb[16:], wgt[16:])
    assert torch.isfinite(out).all()
    assert rel(torch.cat([o_a, o_b]), out) < 1e-5
    scale = max(float(g.norm()) for g in g32.values())
    worst, worst_scalar = ("", 0.0), ("", 0.0)
    for n, g in g32.items():
        s = g_a[n] + g_b[n]
        err = float((g - s).norm()) / (float(s.norm()) + 1e-4 * scale)
        if g.numel() <= 12:  # scalar PReLU slopes: one signed fp32 sum over ~1e8 activations on both sides
            worst_scalar = max(worst_scalar, (n, err), key=lambda kv: kv[1])
        else:
            worst = max(worst, (n, err), key=lambda kv: kv[1])
    print("worst tensor:", worst, " worst scalar slope:", worst_scalar)
    assert worst[1] < 3e-3, worst
    assert worst_scalar[1] < SCALAR_TOL, worst_scalar


def test_sru_scale_x_other_than_one_end_to_end():
    """every one of the 8 SRU layers' `scale_x` buffers set to a different value != 1 in the STATE DICT: the k = 3 skip input x * scale_x
    (layers 1-3; layer 0 has its own skip projection and ignores it, as sru does) must reach the kernels through PreparedWeights.
    Against the reference's waveform and its float64 gradient of one SRU weight (tests/golden/scale_x.npz, oracle/gen_golden_long.py, which
    also checks that the buffers move the waveform by > 1e-2: the test is not vacuous)."""
    import numpy as np

    from oracle.gen_golden_long import SCALE_X_GRAD, scale_x_state
    from util import load_npz

    z = load_npz("scale_x.npz")
    model, sd, cfg = make_model(2, "cuda")
    sd = scale_x_state(sd)
    model.load_state_dict(sd)  # also drops the cached kernel-layout weights (post-load hook)
    mix, _, emb = synth.synth_inputs(2, 16000, 25)
    assert np.array_equal(mix[:, :256].numpy(), z["mix_head"])
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
    assert rel(out, torch.from_numpy(z["out"])) < 1e-3
    # and through the training-step path
    model.zero_grad(set_to_none=True)
    wgt = torch.randn(2, 1, 16000, generator=torch.Generator().manual_seed(5))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        (model(mix.cuda(), emb.cuda()) * wgt.cuda()).sum().backward()
    assert rel(dict(model.named_parameters())[SCALE_X_GRAD].grad, torch.from_numpy(z["grad"])) < 3e-3


def test_weight_cache_invalidation():
    """writes through `.data` do not bump the version counters the prepared-weight cache is keyed on -> explicit
    invalidate_hip_cache(); load_state_dict / train() / eval() invalidate by themselves."""
    model, sd, _ = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(1, 8000, 12)
    mix, emb = mix.cuda(), emb.cuda()
    with torch.no_grad():
        a = model(mix, emb)
        w = model.mask_generator.mask_generator[1].full_layer[2].weight
        w.data.mul_(0.5)
        model.invalidate_hip_cache()
        b = model(mix, emb)
        assert rel(b, a) > 1e-3
        model.load_state_dict(sd)  # post-load hook
        c = model(mix, emb)
        assert torch.equal(c, a)


def test_model_on_non_default_device_index_guard():
    """lib.call launches on the device that owns the tensors and refuses mixed-device arguments; with one GPU only the
    refusal and the explicit-device path can be exercised"""
    from rtfs_net_amd import lib

    x = torch.zeros(4, 256, device="cuda:0")
    with pytest.raises(ValueError):
        lib.call("rtfs_stft_fwd", x, torch.zeros(8), 1, 1024)  # CPU tensor among the arguments
    with torch.cuda.device(0):
        wav = torch.randn(1, 4096, device="cuda:0")
        spec = torch.empty(1 * 33 * 129 * 2, device="cuda:0")
        lib.call("rtfs_stft_fwd", wav, spec, 1, 4096)
    torch.cuda.synchronize()
    assert torch.isfinite(spec).all()
