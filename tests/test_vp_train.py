"""GPU: the VP (video) block's training step on HIP kernels (csrc/vp_train.hip, models/vp_train.py) against the PyTorch-glue module
(models/modules.py TDANetBlock, the torch execution of separators/tdanet.py:106-133): forward, input gradient and EVERY parameter
gradient, with BatchNorm1d batch statistics (train mode, dropout off so both runs see the same function) and with running statistics
(eval mode under autograd); running-statistics update included.  Tolerance 2e-4 relative L2 per tensor (fp32 on both sides)."""
import copy

import pytest
import torch

from util import make_model, rel

pytestmark = pytest.mark.gpu


def _block(train):
    model, _, _ = make_model(2, "cuda")
    vb = model.refinement_module.video_net.get_block(0)
    for mod in vb.modules():  # dropout / DropPath off: the HIP and glue runs must see the same function
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    vb.train(train)
    return vb


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,Tv", [(3, 50), (2, 25), (4, 13), (1, 100), (2, 230)])
def test_vp_block_training_step_matches_the_glue(train, B, Tv):
    from rtfs_net_amd.models.vp_train import VPTrainer, supported, vp_block_train

    vb = _block(train)
    assert supported(vb)
    ref = copy.deepcopy(vb)
    g = torch.Generator().manual_seed(Tv)
    x = torch.randn(B, 512, Tv, generator=g).cuda()
    wgt = torch.randn(B, 512, Tv, generator=g).cuda()
    x1 = x.clone().requires_grad_(True)
    out = vp_block_train(VPTrainer(vb), x1)
    (out * wgt).sum().backward()
    x2 = x.clone().requires_grad_(True)
    out_ref = ref(x2)
    (out_ref * wgt).sum().backward()
    torch.cuda.synchronize()
    assert rel(out, out_ref) < 2e-5
    assert rel(x1.grad, x2.grad) < 2e-4
    worst = ("", 0.0)
    scale = max(float(p.grad.norm()) for p in ref.parameters())
    for (n, p), (_, q) in zip(vb.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        e = float((p.grad - q.grad).norm()) / (float(q.grad.norm()) + 1e-4 * scale)  # (conv biases in front of a batch-statistics BatchNorm: analytically zero)
        worst = max(worst, (n, e), key=lambda kv: kv[1])
    print("worst parameter gradient:", worst)
    assert worst[1] < (2e-3 if train else 2e-4), worst  # (train mode: 26 chained batch normalisations on B * T <= 150 positions amplify fp32 round-off)
    if train:  # running statistics of all 26 BatchNorm layers
        for (n, b1), (_, b2) in zip(vb.named_buffers(), ref.named_buffers()):
            if n.endswith("running_mean") or n.endswith("running_var"):
                assert rel(b1, b2) < 1e-4, n
            if n.endswith("num_batches_tracked"):
                assert int(b1) == int(b2), n


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,Tv", [(3, 50), (2, 25), (2, 12), (1, 100), (1, 230)])
def test_vp_block_training_step_matches_the_reference(train, B, Tv):
    """the same step against float64 autograd of THE REFERENCE's video TDANetBlock (tests/golden/vpgrads_*.npz, written by
    oracle/gen_golden_grads.py from /root/reference, which also holds the oracle's tdanet_block to 1e-7 of it): output, input gradient, every
    parameter gradient, and in train mode the 26 running means / variances.  The fixture's inputs were moved out of round-off distance of
    the PReLU / ReLU kinks first (oracle/regimes.py stable_emb)."""

    from rtfs_net_amd.models.vp_train import VPTrainer, vp_block_train
    from util import load_npz

    z = load_npz(f"vpgrads_{'train' if train else 'eval'}_B{B}_Tv{Tv}.npz")
    vb = _block(train)
    x, wgt = torch.from_numpy(z["x"]), torch.from_numpy(z["wgt"])
    assert x.shape == (B, 512, Tv)
    x1 = x.cuda().requires_grad_(True)
    out = vp_block_train(VPTrainer(vb), x1)
    (out * wgt.cuda()).sum().backward()
    assert rel(out, torch.from_numpy(z["out"])) < 2e-5
    assert rel(x1.grad, torch.from_numpy(z["dx"])) < 1e-3
    ref = {k[5:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("grad.")}
    scale = max(float(v.norm()) for v in ref.values())
    worst = ("", 0.0)
    for n, p in vb.named_parameters():
        r = ref[n]
        assert p.grad is not None, n
        if float(r.norm()) < 1e-6 * scale:
            assert float(p.grad.norm()) < 1e-4 * scale, n  # (conv bias in front of a batch-statistics BatchNorm)
            continue
        e = float((p.grad.double().cpu() - r).norm()) / (float(r.norm()) + 1e-4 * scale)
        worst = max(worst, (n, e), key=lambda kv: kv[1])
    print("worst parameter gradient vs the reference:", worst)
    assert worst[1] < 3e-3, worst
    if train:
        stats = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("stat.")}
        assert len(stats) == 52
        for n, b in vb.named_buffers():
            if n.endswith(("running_mean", "running_var")):
                assert rel(b, stats[n]) < 1e-4, n


def _global_attention_with_masks(ga, g, masks):
    """torch restatement of MultiHeadSelfAttention.forward (layers/attention.py:57-73) + FeedForwardNetwork.forward (conv_layers.py:250-257) with the
    stochastic layers' keep-masks of one step injected (layout of rtfs_net_amd.models.vp_train.attn_masks)"""
    B, C, Tg = g.shape
    na, ne = 8 * Tg * Tg, Tg * 64
    m_attn, m_el, dp = masks[:, :na].view(B, 8, Tg, Tg), masks[:, na:na + ne].view(B, Tg, 64), masks[:, na + ne:]
    m, f = ga.MHSA, ga.FFN
    y = m.norm1(g.transpose(1, 2)) + m.pos_enc.pe[:, :Tg]
    qkv = y @ m.attention.in_proj_weight.t() + m.attention.in_proj_bias
    q, k, v = (t.view(B, Tg, 8, 8).transpose(1, 2) for t in qkv.split(64, dim=-1))
    p = torch.softmax(q @ k.transpose(-1, -2) / 8 ** 0.5, -1) * m_attn
    o = (p @ v).transpose(1, 2).reshape(B, Tg, 64)
    a = o @ m.attention.out_proj.weight.t() + m.attention.out_proj.bias
    x1 = m.norm2(a * m_el + y).transpose(1, 2) * dp[:, 0].view(B, 1, 1) + g
    r = f.refiner(f.encoder(x1)) * dp[:, 1].view(B, 1, 1)
    return f.decoder(r) * dp[:, 2].view(B, 1, 1) + x1


@pytest.mark.parametrize("B,Tg,drop", [(3, 7, True), (2, 13, True), (4, 4, False), (1, 16, True), (2, 2, False)])
def test_global_attention_hip_training_kernels(B, Tg, drop):
    """csrc/vp_attn.hip (rtfs_vp_attn_fwd / _bwd through VPAttnFn) against torch autograd: output, input gradient and all 16 parameter
    gradients, with one explicit draw of the dropout / DropPath keep-masks (and, without masks, against the module itself in eval mode)."""
    from rtfs_net_amd.models import vp_train as vt

    vb = _block(True)
    ga = vb.globalatt[0]
    assert vt.attn_supported(ga)
    gen = torch.Generator().manual_seed(10 * B + Tg)
    g = torch.randn(B, 64, Tg, generator=gen).cuda()
    wgt = torch.randn(B, 64, Tg, generator=gen).cuda()
    masks = None
    if drop:
        for mod in ga.modules():  # restore the stochastic layers (the helper switched them off)
            if isinstance(getattr(mod, "p", None), float):
                mod.p = 0.1
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.1
        ga.FFN.dropout_layer.p = 0.3  # (a per-utterance factor: make a drop likely at these batch sizes)
        torch.manual_seed(5)
        masks = vt.attn_masks(ga, B, Tg, g.device)
        assert masks is not None and masks.shape == (B, 8 * Tg * Tg + 64 * Tg + 3)
        keep = masks[:, :8 * Tg * Tg]
        assert bool(((keep == 0) | ((keep - 1 / 0.9).abs() < 1e-6)).all()) and 0.02 < float((keep == 0).float().mean()) < 0.25
    params = vt.attn_params(ga)
    g1 = g.clone().requires_grad_(True)
    out = vt.VPAttnFn.apply(ga, masks, g1, *params)
    (out * wgt).sum().backward()
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    g2 = g.clone().requires_grad_(True)
    if drop:
        ref = _global_attention_with_masks(ga, g2, masks)
    else:
        ga.eval()
        ref = ga(g2)
    (ref * wgt).sum().backward()
    torch.cuda.synchronize()
    assert rel(out, ref) < 2e-6
    assert rel(g1.grad, g2.grad) < 2e-5
    scale = max(float(p.grad.norm()) for p in params)
    for i, p in enumerate(params):
        e = float((got[i] - p.grad).norm()) / (float(p.grad.norm()) + 1e-5 * scale)
        assert e < 1e-4, (i, tuple(p.shape), e)


@pytest.mark.parametrize("B,Tg", [(2, 17), (3, 27), (1, 188), (2, 64), (1, 500)])
def test_global_attention_hip_eval_more_than_16_tokens(B, Tg):
    """rtfs_vp_attn_long_fwd (csrc/vp_attn.hip vp_attn_long_fwd_kernel: GlobalAttention in eval mode with its per-token intermediates in a global
    workspace, utterances longer than 5.1 s: 27 pooled tokens at 8.5 s, 188 at 120 s) against the module itself (attention.py:28-73,
    conv_layers.py:218-259), and at 16 tokens against the one-workgroup LDS kernel."""
    from rtfs_net_amd import lib
    from rtfs_net_amd.models import vp_train as vt

    vb = _block(True)
    ga = vb.globalatt[0].eval()
    assert vt.attn_supported(ga)
    gen = torch.Generator().manual_seed(7 * B + Tg)
    g = torch.randn(B, 64, Tg, generator=gen).cuda()
    packed = torch.cat([p.detach().float().reshape(-1) for p in vt.attn_params(ga)])
    pe = ga.MHSA.pos_enc.pe[0, :Tg].float().contiguous()
    out = torch.full_like(g, float("nan"))
    work = torch.empty(B * lib.load().rtfs_vp_attn_long_work_floats(Tg), device="cuda")
    lib.call("rtfs_vp_attn_long_fwd", g, packed, pe, out, work, B, Tg)
    with torch.no_grad():
        ref = ga(g)
    assert rel(out, ref) < 2e-6
    g16 = g[:, :, :16].contiguous()
    a, b = torch.empty_like(g16), torch.empty_like(g16)
    lib.call("rtfs_vp_attn_fwd", g16, packed, pe[:16].contiguous(), None, a, B, 16)
    lib.call("rtfs_vp_attn_long_fwd", g16, packed, pe[:16].contiguous(), b, work, B, 16)
    assert rel(b, a) < 1e-6
    with pytest.raises(RuntimeError):
        lib.call("rtfs_vp_attn_long_fwd", g, packed, pe, out, work, B, 1025)


@pytest.mark.parametrize("B,Tv", [(3, 50), (2, 12), (1, 100), (2, 7), (1, 230)])
def test_caf_video_side_hip_training_kernels(B, Tv):
    """CAFVideoFn (rtfs_caf_video_fwd + rtfs_caf_video_bwd) against torch autograd over the cell's modules (layers/fusion.py:255,262-265 as the
    glue path of round 2 ran it): att, rsz, the input gradient and the eight parameter gradients"""
    from rtfs_net_amd.models.vp_train import caf_video_train

    model, _, _ = make_model(2, "cuda")
    cell = model.refinement_module.crossmodal_fusion.get_fusion_block(0).audio_lstm
    gen = torch.Generator().manual_seed(Tv)
    v = torch.randn(B, 512, Tv, generator=gen).cuda()
    wa, wr = torch.randn(B, Tv, 256, generator=gen).cuda(), torch.randn(B, Tv, 256, generator=gen).cuda()
    params = [p for n, p in cell.named_parameters() if n.startswith(("attention_embed.", "resize."))]
    assert len(params) == 8
    v1 = v.clone().requires_grad_(True)
    att, rsz = caf_video_train(cell, v1)
    ((att * wa).sum() * 50 + (rsz * wr).sum()).backward()  # (softmax outputs are ~1/Tv: weight them up)
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    v2 = v.clone().requires_grad_(True)
    a = torch.softmax(cell.attention_embed(v2).reshape(B, 256, 4, -1).mean(2), -1).transpose(1, 2)
    r = cell.resize(v2).transpose(1, 2)
    ((a * wa).sum() * 50 + (r * wr).sum()).backward()
    torch.cuda.synchronize()
    assert rel(att, a) < 2e-6 and rel(rsz, r) < 2e-6
    assert rel(v1.grad, v2.grad) < 5e-5
    scale = max(float(p.grad.norm()) for p in params)
    for g, p in zip(got, params):
        # (the attention embedding's gLN shift is analytically gradient-free - a per-channel constant cancels in the softmax over Tv: both sides
        # hold fp32 residue there, hence the absolute floor)
        e = float((g - p.grad).norm()) / (float(p.grad.norm()) + 1e-3 * scale)
        assert e < 2e-4, (tuple(p.shape), e)


@pytest.mark.parametrize("B,Tv", [(2, 120), (1, 230), (1, 1000)])
def test_vp_block_eval_chain_for_long_inputs(B, Tv):
    """inputs longer than the one-kernel inference form holds (Tv > 100 = 4 s): the multi-launch kernels with running statistics
    (models/vp_train.py vp_block_eval; GlobalAttention on HIP for Tv = 120 (15 pooled tokens), the module beyond) against the oracle's block"""
    from oracle import avnet_ref
    from rtfs_net_amd.models.vp_train import vp_block_eval
    from util import VIDEO_PREFIX

    model, sd, cfg = make_model(2, "cuda")
    vb = model.refinement_module.video_net.get_block(0).eval()
    x = torch.randn(B, 512, Tv, generator=torch.Generator().manual_seed(Tv))
    out = vp_block_eval(vb, x.cuda())
    with torch.no_grad():
        ref = avnet_ref.tdanet_block(x, avnet_ref.P(sd).sub(VIDEO_PREFIX), avnet_ref.normalise_cfg(cfg)["video"])
    assert rel(out, ref) < 2e-5
