"""rtfs_net_amd.optim.FusedAdamW (csrc/optim.hip: gradient clipping + AdamW of the training step as two launches) against
torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW - the pair the reference's trainer runs (config yaml:117-120, train.py:135-146)."""
import copy

import pytest
import torch

from util import make_model, rel


def _close(a, b, lr=1e-3):
    """error of an updated parameter relative to max(its norm, the size of one update): a scalar parameter that an update of size lr moves to 1e-5 carries the
    update's rounding 100 times magnified if it is measured against its own value"""
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / max(float(b.norm()), lr * b.numel() ** 0.5))


def test_fused_adamw_is_an_optimizer_with_adamw_state_keys():
    from rtfs_net_amd.optim import FusedAdamW

    w = torch.nn.Parameter(torch.zeros(3))
    opt = FusedAdamW([w], lr=1e-3, weight_decay=0.1)
    assert isinstance(opt, torch.optim.Optimizer) and opt.param_groups[0]["betas"] == (0.9, 0.999) and opt.param_groups[0]["eps"] == 1e-8
    assert opt.step() is None  # no gradient: nothing to do (and no GPU needed)
    with pytest.raises(ValueError):
        FusedAdamW([w], lr=-1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("max_norm", [5.0, 1e-2, None])
def test_fused_adamw_matches_torch_adamw_and_clip(max_norm):
    """three steps on RTFS-Net-2's 403 parameter tensors with random gradients (clipping inactive at 5.0 on step 0 or not, always active at 1e-2, off):
    parameters, both moments and the clipped gradients against the torch pair; optimizer state_dicts interchange"""
    from rtfs_net_amd.optim import FusedAdamW

    model, _, _ = make_model(2, "cuda")
    ref = copy.deepcopy(model)
    pa, pb = list(model.parameters()), list(ref.parameters())
    fused = FusedAdamW(pa, lr=1e-3, weight_decay=0.1)
    torch_opt = torch.optim.AdamW(pb, lr=1e-3, weight_decay=0.1)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for it in range(3):
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, device="cuda", generator=gen) * (0.3 if it else 3.0)
            a.grad, b.grad = g.clone(), g.clone()
        fused.step(max_norm=max_norm)
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(pb, max_norm)
        torch_opt.step()
        torch.cuda.synchronize()
        worst = max(_close(a, b) for a, b in zip(pa, pb))
        worst_g = max(rel(a.grad, b.grad) for a, b in zip(pa, pb))
        assert worst < 1e-6 and worst_g < 1e-6, (it, worst, worst_g)
    for a, b in zip(pa, pb):
        sa, sb = fused.state[a], torch_opt.state[b]
        assert float(sa["step"]) == float(sb["step"]) == 3.0
        assert _close(sa["exp_avg"], sb["exp_avg"], 0.1) < 1e-6 and _close(sa["exp_avg_sq"], sb["exp_avg_sq"], 1e-3) < 1e-6  # (scales of one moment update)
    # the state interchanges with torch.optim.AdamW's (same keys): continue the torch optimizer's run in the fused one and vice versa
    fused2 = FusedAdamW(pa, lr=1e-3, weight_decay=0.1)
    fused2.load_state_dict(torch_opt.state_dict())
    torch2 = torch.optim.AdamW(pb, lr=1e-3, weight_decay=0.1)
    torch2.load_state_dict(fused.state_dict())
    for a, b in zip(pa, pb):
        g = torch.randn(a.shape, device="cuda", generator=gen)
        a.grad, b.grad = g.clone(), g.clone()
    fused2.step(max_norm=max_norm)
    if max_norm is not None:
        torch.nn.utils.clip_grad_norm_(pb, max_norm)
    torch2.step()
    assert max(_close(a, b) for a, b in zip(pa, pb)) < 1e-6
    assert float(fused2.state[pa[0]]["step"]) == 4.0


@pytest.mark.gpu
def test_fused_adamw_survives_a_host_that_runs_ahead():
    """the gradient pointer row of a step is staged in pinned memory and copied asynchronously: many steps enqueued without a synchronisation must each
    see THEIR gradients (a reused staging row would be overwritten by the host before the device has copied it)"""
    from rtfs_net_amd.optim import FusedAdamW

    p = [torch.nn.Parameter(torch.zeros(1 << 20, device="cuda")) for _ in range(4)]
    q = [torch.nn.Parameter(torch.zeros(1 << 20, device="cuda")) for _ in range(4)]
    fused, ref = FusedAdamW(p, lr=1e-2, weight_decay=0.0), torch.optim.AdamW(q, lr=1e-2, weight_decay=0.0)
    big = torch.randn(4096, 4096, device="cuda")
    for it in range(40):
        (big @ big).sum()  # keep the device busy so that the host gets ahead
        for a, b in zip(p, q):
            g = torch.full((1 << 20,), float(it % 5) - 2.0, device="cuda")
            a.grad, b.grad = g, g.clone()
        fused.step()
        ref.step()
    torch.cuda.synchronize()
    assert max(_close(a, b, 1e-2) for a, b in zip(p, q)) < 1e-6


@pytest.mark.gpu
def test_fused_adamw_step_reaches_the_hip_forward():
    """the HIP path keeps kernel-layout copies of the weights keyed on the parameters' version counters; FusedAdamW writes the parameters through raw
    pointers and must move those counters - two training steps with it give the outputs of two steps with torch.optim.AdamW + clip_grad_norm_"""
    from rtfs_net_amd.optim import FusedAdamW
    from util import synth

    outs = []
    for fused in (True, False):
        model, _, _ = make_model(2, "cuda")
        model.train()
        for mod in model.modules():  # no stochastic layers: the two runs must see the same forward
            if isinstance(getattr(mod, "p", None), float):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        mix, _, emb = synth.synth_inputs(2, 8000, 12)
        mix, emb = mix.cuda(), emb.cuda()
        opt = FusedAdamW(model.parameters(), lr=1e-2, weight_decay=0.1) if fused else torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.1)
        v0 = next(model.parameters())._version
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            model(mix, emb).square().mean().backward()
            if fused:
                opt.step(max_norm=5.0)
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
                opt.step()
        assert next(model.parameters())._version > v0
        with torch.no_grad():  # a third forward, still in train mode: the biases in front of BatchNorm layers have analytically zero gradients, Adam turns their
            outs.append(model(mix, emb).clone())  # rounding noise into +-lr updates that differ run to run and only show through RUNNING statistics (eval mode)
    first = make_model(2, "cuda")[0].train()
    for mod in first.modules():
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    with torch.no_grad():
        assert rel(outs[0], first(mix, emb)) > 1e-3  # the two steps did move the output
    assert rel(outs[0], outs[1]) < 1e-3  # (observed 3e-5 ... 3e-4 at lr = 1e-2, run to run: the noise-driven +-lr updates of the zero-gradient parameters are not exactly invisible in fp32; stale weights fail the line above)


def _same_tree(a, b, path=""):
    """nested dicts / lists / tuples of tensors, floats, None: identical structure and values"""
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), (path, sorted(set(a) ^ set(b)))
        for k in a:
            _same_tree(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert isinstance(b, (list, tuple)) and len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same_tree(x, y, f"{path}[{i}]")
    elif isinstance(a, torch.Tensor):
        assert isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), path
    else:
        assert a == b, (path, a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [3, 1])
def test_gather_train_weights_packed_modes(prec):
    """the same in the bf16 / split-bf16 modes: the host-packed MFMA weights (hip_path.pack_bf16) come out of ONE packing pass over a contiguous region of
    the gathered fp32 layouts - bit for bit what TrainWeights packs tensor by tensor, before and after the parameters move"""
    from rtfs_net_amd.models.hip_train import GatherTrainWeights, TrainWeights

    model, _, _ = make_model(2, "cuda")
    model.train()

    def rebuilt():
        t = TrainWeights(model, prec)
        for k in ("caf_key_s", "caf_key_b", "caf_value_s", "caf_value_b"):
            t.w.pop(k)
        return t

    gw = GatherTrainWeights(model, prec)
    for _ in range(2):
        tw = rebuilt()
        assert gw.blocks[0]["pw"].dtype == torch.bfloat16 and gw.blocks[0]["dp0"]["layers"][1]["w"].dtype == torch.float32
        for a, b in ((gw.w, tw.w), (gw.blocks, tw.blocks), (gw._scal, tw._scal)):
            _same_tree(a, b)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.01)
        gw.refresh(model)


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
def test_gather_train_weights_equal_the_rebuilt_ones(training):
    """GatherTrainWeights (the kernel-layout weight copies of the training step as one multi-tensor copy + one gather, plan built by running TrainWeights'
    construction on index-valued stand-ins) against TrainWeights itself: every tensor bit for bit, every scalar slot, before and after the parameters move
    (in-place updates, as an optimizer step makes them), in train mode and in eval mode under autograd (CAF BatchNorm folded with its running statistics)"""
    from rtfs_net_amd.models.hip_train import GatherTrainWeights, TrainWeights

    model, _, _ = make_model(3, "cuda")
    model.train(training)
    def rebuilt():
        t = TrainWeights(model, 0)
        if training:  # the CAF embeddings' BatchNorm folded with RUNNING statistics: only read in eval mode, where the gather form recomputes it
            for k in ("caf_key_s", "caf_key_b", "caf_value_s", "caf_value_b"):
                t.w.pop(k)
        return t

    gw = GatherTrainWeights(model, 0)
    tw = rebuilt()
    for a, b in ((gw.w, tw.w), (gw.blocks, tw.blocks), (gw._scal, tw._scal)):
        _same_tree(a, b)
    assert gw.version == tw.version and gw.caf_prefix == tw.caf_prefix
    gen = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, device="cuda", generator=gen) * 0.05)
        for n, b in model.named_buffers():
            if n.endswith(("running_mean", "running_var", "scale_x")):
                b.add_(torch.rand(b.shape, device="cuda", generator=gen) * 0.05)
    assert gw.same_storage(model) and gw.version != TrainWeights.fingerprint(model, training=training)
    held = gw.blocks[0]["pw"]  # the nested dicts are persistent views: a reference taken before the refresh sees the new values
    gw.refresh(model)
    tw = rebuilt()
    for a, b in ((gw.w, tw.w), (gw.blocks, tw.blocks), (gw._scal, tw._scal)):
        _same_tree(a, b)
    assert gw.version == tw.version and torch.equal(held, tw.blocks[0]["pw"])
    model.cpu()
    model.cuda()
    assert not gw.same_storage(model)  # re-allocated parameters: HipTrainer.weights() builds a new plan


@pytest.mark.gpu
def test_caf_batchnorm_kernels_match_the_torch_arithmetic():
    """rtfs_caf_bn_prepare / rtfs_caf_bn_adjoint (csrc/optim.hip: the per-channel BatchNorm2d arithmetic of the CAF cell's key / value embeddings in the training
    step, fusion.py:249-253, as one launch each) against the torch-op form they replace (model._hip.fuse["cafbn"] = False): the waveform, every gradient of the
    cell's embeddings, the running statistics and num_batches_tracked after one step"""
    from util import synth

    res = []
    for fused in (True, False):
        model, _, _ = make_model(2, "cuda")
        model.train()
        for mod in model.modules():
            if isinstance(getattr(mod, "p", None), float):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        model._hip.fuse["cafbn"] = fused
        mix, _, emb = synth.synth_inputs(3, 8000, 12)
        out = model(mix.cuda(), emb.cuda())
        out.square().mean().backward()
        cell = model.refinement_module.crossmodal_fusion.get_fusion_block(0).audio_lstm
        named = dict(cell.named_parameters())
        grads = {n: p.grad.clone() for n, p in named.items() if n.startswith(("key_embed", "value_embed"))}
        stats = {n: b.clone() for n, b in cell.named_buffers() if n.startswith(("key_embed", "value_embed"))}
        res.append((out.detach().clone(), grads, stats))
    (oa, ga, sa), (ob, gb, sb) = res
    assert rel(oa, ob) < 1e-6 and len(ga) == 6 and len(sa) == 6
    for n in ga:
        assert rel(ga[n], gb[n]) < 1e-4, (n, rel(ga[n], gb[n]))  # (d(dw) is a difference of three sums of ~1e5 terms each: 2e-5 between two fp32 evaluation orders)
    for n in sa:
        if n.endswith("num_batches_tracked"):
            assert int(sa[n]) == int(sb[n]) == 1
        else:
            assert rel(sa[n], sb[n]) < 1e-6, n


@pytest.mark.gpu
def test_backward_after_a_newer_forward_relaid_the_weights_is_refused():
    """GatherTrainWeights refreshes its kernel-layout buffers in place.  forward(A) -> parameter update -> forward(B) -> backward(A) would run A's
    adjoint on B's weights; the step refuses loudly.  Without a parameter change in between (gradient accumulation) both orders are fine."""
    from util import synth

    model, _, _ = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(1, 8000, 12)
    mix, emb = mix.cuda(), emb.cuda()
    a = model(mix, emb).square().mean()
    b = model(mix, emb).square().mean()  # same parameters: nothing is re-laid out
    b.backward(), a.backward()
    a = model(mix, emb).square().mean()
    with torch.no_grad():
        next(model.parameters()).mul_(1.0)  # (moves the version counter)
    model(mix, emb)
    with pytest.raises(RuntimeError, match="parameters changed"):
        a.backward()


@pytest.mark.gpu
def test_fused_adamw_mixed_step_counts_nan_norm_and_moved_storage():
    """parameters whose step counts differ get their own bias corrections (torch.optim.AdamW corrects per parameter); a NaN total norm reaches
    every gradient as torch's clip does; a parameter whose storage moved under the same Parameter object is followed."""
    from rtfs_net_amd.optim import FusedAdamW

    gen = torch.Generator(device="cuda").manual_seed(5)
    pa = [torch.nn.Parameter(torch.randn(n, device="cuda", generator=gen)) for n in (7, 1500, 64)]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    fused, ref = FusedAdamW(pa, lr=1e-2, weight_decay=0.1), torch.optim.AdamW(pb, lr=1e-2, weight_decay=0.1)
    for it in range(4):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 1 and it < 2:  # parameter 1 receives its first gradient two steps late
                a.grad = b.grad = None
                continue
            g = torch.randn(a.shape, device="cuda", generator=gen)
            a.grad, b.grad = g.clone(), g.clone()
        if it == 3:  # the storage of parameter 2 moves (what module.to() / p.data = ... do)
            pa[2].data = pa[2].data.clone()
        fused.step(max_norm=1.0)
        torch.nn.utils.clip_grad_norm_([p for p in pb if p.grad is not None], 1.0)
        ref.step()
        assert max(_close(a, b, 1e-2) for a, b in zip(pa, pb)) < 1e-6, it
    assert [float(fused.state[p]["step"]) for p in pa] == [4.0, 2.0, 4.0]
    for a in pa:
        a.grad = torch.ones_like(a)
    pa[0].grad[0] = float("nan")
    fused.step(max_norm=1.0)
    assert all(bool(torch.isnan(a.grad).all()) and bool(torch.isnan(a).all()) for a in pa)
