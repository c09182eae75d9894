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
@pytest.mark.parametrize("B,Tv", [(3, 50), (2, 25), (4, 13), (1, 100)])
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
        e = float((p.grad - q.grad).norm()) / (float(q.grad.norm()) + 1e-5 * scale)
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
@pytest.mark.parametrize("B,Tv", [(3, 50), (2, 25), (2, 12), (1, 100)])
def test_vp_block_training_step_matches_the_oracle(train, B, Tv):
    """the same step against float64 autograd of the ORACLE's TDANetBlock restatement (oracle/avnet_ref.py tdanet_block, pinned to the
    reference by tests/golden): output, input gradient, every parameter gradient, and in train mode the 26 running means / variances
    (F.batch_norm(training=True) updates the oracle's copies in place).  Inputs are moved out of round-off distance of the PReLU / ReLU
    kinks first (util.stable_emb)."""
    from oracle import avnet_ref
    from rtfs_net_amd.models.vp_train import VPTrainer, vp_block_train
    from util import VIDEO_PREFIX, stable_emb

    model, sd, cfg = make_model(2, "cuda")
    vb = model.refinement_module.video_net.get_block(0)
    for mod in vb.modules():
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    vb.train(train)
    g = torch.Generator().manual_seed(100 + Tv)
    x = stable_emb(sd, cfg, torch.randn(B, 512, Tv, generator=g), train)
    wgt = torch.randn(B, 512, Tv, generator=g)
    x1 = x.cuda().requires_grad_(True)
    out = vp_block_train(VPTrainer(vb), x1)
    (out * wgt.cuda()).sum().backward()
    nograd = ("running_mean", "running_var", ".pe", "num_batches_tracked")
    sd64 = {k: (v.double().clone().requires_grad_(not k.endswith(nograd)) if v.is_floating_point() else v.clone()) for k, v in sd.items()
            if k.startswith(VIDEO_PREFIX)}
    x64 = x.double().requires_grad_(True)
    ref_out = avnet_ref.tdanet_block(x64, avnet_ref.P(sd64).sub(VIDEO_PREFIX), avnet_ref.normalise_cfg(cfg)["video"], training=train)
    (ref_out * wgt.double()).sum().backward()
    assert rel(out, ref_out) < 2e-5
    assert rel(x1.grad, x64.grad) < 1e-3
    scale = max(float(v.grad.norm()) for v in sd64.values() if v.is_floating_point() and v.grad is not None)
    worst = ("", 0.0)
    for n, p in vb.named_parameters():
        r = sd64[f"{VIDEO_PREFIX}.{n}"].grad
        assert p.grad is not None and r is not None, n
        if float(r.norm()) < 1e-6 * scale:
            assert float(p.grad.norm()) < 1e-4 * scale, n  # (conv bias in front of a batch-statistics BatchNorm)
            continue
        e = float((p.grad.double().cpu() - r).norm()) / (float(r.norm()) + 1e-4 * scale)
        worst = max(worst, (n, e), key=lambda kv: kv[1])
    print("worst parameter gradient vs oracle:", worst)
    assert worst[1] < 3e-3, worst
    if train:
        for n, b in vb.named_buffers():
            if n.endswith(("running_mean", "running_var")):
                assert rel(b, sd64[f"{VIDEO_PREFIX}.{n}"]) < 1e-4, n
