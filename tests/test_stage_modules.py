"""GPU: the reference's MODULE PROTOCOL on the build's AVNet.

The reference's AVNet is five ordinary modules called in sequence (src/models/tdavnet.py:86-97); its `get_MACs` calls them one by one
(TDAVNet/base_av_model.py:61-118) and the fixture generator captures every stage with forward hooks (oracle/gen_golden.py `run_reference`).
Here the SAME hook list - same attribute paths, same naming of repeated calls - runs on rtfs_net_amd's AVNet and is compared with what it
captured on the reference itself (tests/golden/rtfs6_b2.npz: 4096-point strided samples + norms of every hook output of all six block applications)."""
import pytest
import torch

from util import load_npz, make_model, rel, synth

pytestmark = pytest.mark.gpu
TOL = 5e-4


def strided(t, n=4096):
    f = t.flatten()
    return f[:: max(1, f.numel() // n)][:n]


def _reference_hook_list(model, taps):
    """oracle/gen_golden.py:60-90, on this model"""

    def hook(name):
        def f(mod, inp, out):
            n, i = name, 0
            while n in taps:  # shared blocks are called several times
                i += 1
                n = f"{name}#{i}"
            taps[n] = out.detach().clone()

        return f

    rm = model.refinement_module
    ab = rm.audio_net.blocks
    return [
        model.encoder.register_forward_hook(hook("a_emb")),
        model.audio_bottleneck.register_forward_hook(hook("a0")),
        ab.register_forward_hook(hook("block")),
        ab.gateway.register_forward_hook(hook("gateway")),
        ab.projection.register_forward_hook(hook("projection")),
        ab.downsample_layers[0].register_forward_hook(hook("down0")),
        ab.downsample_layers[1].register_forward_hook(hook("down1")),
        ab.globalatt[0].register_forward_hook(hook("dp_freq")),
        ab.globalatt[1].register_forward_hook(hook("dp_time")),
        ab.globalatt[2].register_forward_hook(hook("attn")),
        ab.fusion_layers[0].register_forward_hook(hook("tfar0")),
        ab.fusion_layers[1].register_forward_hook(hook("tfar1")),
        ab.concat_layers[0].register_forward_hook(hook("concat0")),
        rm.video_net.blocks.register_forward_hook(hook("vp")),
        rm.crossmodal_fusion.fusion_module.audio_lstm.register_forward_hook(hook("caf")),
        model.mask_generator.register_forward_hook(hook("masked")),
    ]


def test_the_reference_hook_list_runs_unchanged_and_matches_the_reference_taps():
    R, B, L = 6, 2, 32000
    z = load_npz("rtfs6_b2.npz")
    model, _, _ = make_model(R, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, 25 * L // 16000)
    mix, emb = mix.cuda(), emb.cuda()
    with torch.no_grad():
        plain = model(mix, emb)
    taps = {}
    hooks = _reference_hook_list(model, taps)
    with torch.no_grad():
        out = model(mix, emb)
    for h in hooks:
        h.remove()
    assert rel(out, torch.from_numpy(z["out"])) < 1e-3 and rel(out, plain) < 1e-5  # (the hooked route is the same kernels + layout views)
    names = [k[4:] for k in z.files if k.startswith("tap.")]
    assert len(names) == 5 + 11 * R and set(names) == set(taps), set(names) ^ set(taps)
    worst = ("", 0.0)
    for k in names:
        v = taps[k].float().cpu()
        v = v.reshape(B, 256, *v.shape[-2:]) if k == "masked" else v  # ([B, n_src, 256, T, F] as the reference returns it)
        e = rel(strided(v), torch.from_numpy(z["tap." + k]))
        n = abs(float(v.double().norm()) / float(z["norm." + k]) - 1)
        worst = max(worst, (k, max(e, n)), key=lambda kv: kv[1])
        assert e < TOL and n < TOL, (k, e, n)
    print("worst hook output:", worst)
    # no hook left: the fused route again (bit-identical to the first call), nothing fires
    with torch.no_grad():
        assert torch.equal(model(mix, emb), plain)
    assert len(taps) == len(names)


def test_stage_modules_called_one_by_one_like_get_macs():
    """base_av_model.py:75-95: encoder -> audio_bottleneck / video_bottleneck -> refinement_module -> mask_generator -> decoder as separate calls,
    shapes as in the reference, the composition equal to AVNet.forward; inner (fused) modules still refuse a direct call, loudly"""
    model, _, _ = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(2, 16000, 25)
    mix, emb = mix.cuda(), emb.cuda()
    with torch.no_grad():
        ref = model(mix, emb)
        a_emb = model.encoder(mix)
        assert a_emb.shape == (2, 256, 126, 129) and rel(model.encoder(mix[:, None]), a_emb) == 0 and model.encoder(mix[0]).shape == (1, 256, 126, 129)
        audio, video = model.audio_bottleneck(a_emb), model.video_bottleneck(emb)
        refined = model.refinement_module(audio, video)
        sep = model.mask_generator(refined, a_emb)
        assert audio.shape == refined.shape == a_emb.shape and sep.shape == (2, 1, 256, 126, 129)
        out = model.decoder(sep, mix.shape)
        assert out.shape == ref.shape == (2, 1, 16000) and rel(out, ref) < 1e-5
        with pytest.raises(RuntimeError, match="fused into the HIP kernels"):
            model.refinement_module.audio_net.blocks.globalatt[0](torch.zeros(1, 64, 62, 64, device="cuda"))
        with pytest.raises(ValueError):
            model.decoder(sep, (2, 12000))
        with pytest.raises(ValueError):
            model.audio_bottleneck(a_emb[:, :, :, :64])  # not 129 bins
        with pytest.raises(ValueError):
            model.mask_generator(refined, a_emb[:1])
    with pytest.raises(NotImplementedError):  # under autograd the stage modules refuse (the training step is one chain behind AVNet.forward)
        model.encoder(mix)


def test_hooks_observe_only_and_unserved_hooks_are_refused():
    model, _, _ = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(1, 8000, 12)
    mix, emb = mix.cuda(), emb.cuda()
    blk = model.refinement_module.audio_net.blocks
    seen = []
    h = blk.residual_conv.register_forward_hook(lambda m, i, o: seen.append((i[0].shape, o.shape)))
    hp = blk.projection.register_forward_pre_hook(lambda m, i: seen.append(("pre", i[0].shape)))
    with torch.no_grad():
        model(mix, emb)
    assert seen == [("pre", (1, 256, 63, 129)), ((1, 64, 63, 129), (1, 256, 63, 129))] * 2  # two applications of the shared block
    with pytest.raises(NotImplementedError, match="inference path"):
        model(mix, emb)  # autograd enabled: the training step refuses hooks
    h.remove(), hp.remove()
    h = blk.gateway.register_forward_hook(lambda m, i, o: o * 2)
    with torch.no_grad(), pytest.raises(NotImplementedError, match="observe"):
        model(mix, emb)
    h.remove()
    inner = blk.gateway.full_layer[2].register_forward_hook(lambda m, i, o: None)
    served = blk.gateway.register_forward_hook(lambda m, i, o: None)
    with torch.no_grad(), pytest.raises(NotImplementedError, match="cannot be served"):
        model(mix, emb)
    inner.remove(), served.remove()


def test_hooked_route_in_a_split_bf16_mode_and_with_non_shared_blocks():
    """the module-by-module route runs the SAME kernels as the fused one: in the bf16x3 mode the hooked forward equals the fused forward of that mode, and with
    `audio_params.shared = False` (tdanet.py:170-181) every block's own modules fire once, in order"""
    import copy

    from rtfs_net_amd import AVNet

    model, _, _ = make_model(3, "cuda")
    mix, _, emb = synth.synth_inputs(2, 16000, 25)
    mix, emb = mix.cuda(), emb.cuda()
    model.set_compute_dtype("bf16x3")
    with torch.no_grad():
        fused = model(mix, emb)
        seen = []
        h = model.refinement_module.audio_net.blocks.register_forward_hook(lambda m, i, o: seen.append(o.shape))
        hooked = model(mix, emb)
        h.remove()
    model.set_compute_dtype("f32")
    assert len(seen) == 3 and rel(hooked, fused) < 1e-5
    cfg = synth.rtfs_audionet(2)
    cfg["audio_params"]["shared"] = False
    ns = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
    ns.load_state_dict(synth.synth_state_dict(ns.state_dict()))
    ns = ns.cuda()
    order = []
    hooks = [ns.refinement_module.audio_net.blocks[i].residual_conv.register_forward_hook(lambda m, inp, o, i=i: order.append((i, tuple(o.shape)))) for i in (0, 1)]
    with torch.no_grad():
        plain = ns._hip(mix, emb)  # (the fused route, hooks ignored)
        out = ns(mix, emb)
    for hk in hooks:
        hk.remove()
    assert order == [(0, (2, 256, 126, 129)), (1, (2, 256, 126, 129))] and rel(out, plain) < 1e-5
