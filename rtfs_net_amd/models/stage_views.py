"""The reference's module protocol on top of the HIP path: stage modules that can be CALLED, forward hooks that FIRE.

The reference's AVNet is five ordinary modules (src/models/tdavnet.py:86-97) and its own tooling uses them as such: `get_MACs` calls
`self.encoder(...)`, `self.audio_bottleneck(...)`, `self.mask_generator(...)` one by one (TDAVNet/base_av_model.py:61-118), the fixture generator
captures every stage with `register_forward_hook` (oracle/gen_golden.py `run_reference`).  The fused forward (models/hip_path.py) runs the whole path
as one chain of channels-last kernels, so until round 6 every stage module's `forward` raised and hooks never fired.

Here (inference: eval() + torch.no_grad(); the training step keeps its one autograd chain and refuses hooks loudly):

  * `StageViews.encoder / bottleneck / refinement / mask / decoder` are the bodies of the five stage modules' `forward`: NCHW tensors at the module
    boundary as in the reference, the HIP entry points of the fused path in between (layout changes, statistics and normalised views included:
    csrc/views.hip) - no torch arithmetic.
  * `AVNet.forward` takes the module-by-module route when any servable module carries a forward (pre-)hook; torch fires the stage modules' hooks
    itself (they are called through `__call__`), and `fire_refinement_hooks` fires the hooks of the modules INSIDE the refinement module - every RTFS
    block application and its gateway / projection / downsample_layers / globalatt[0..2] / fusion_layers / concat_layers / residual_conv, the VP block,
    the CAF cell - in the reference's execution order, with the module's inputs and output materialised in NCHW from the stage taps of the forward.
    Hooks observe: one that returns a replacement output is refused (the kernels downstream have already run).
"""
from __future__ import annotations

import torch

from .hip_path import C, F2, F_BINS, H, lib  # (the ctypes binding as hip_path resolved it: the models package is relocatable, train.py:95)


def _nchw(x_cl, B, Cc, T, Fq):
    out = torch.empty(B, Cc, T, Fq, device=x_cl.device)
    lib.call("rtfs_cl_to_nchw", x_cl, out, B, T * Fq, Cc)
    return out


def _cl(x_nchw):
    x = x_nchw.to(torch.float32).contiguous()
    B, Cc, T, Fq = x.shape
    out = torch.empty(B * T * Fq * Cc, device=x.device)
    lib.call("rtfs_nchw_to_cl", x, out, B, T * Fq, Cc)
    return out


def _check_map(x, what, chan=C):
    """the stage modules take the reference's [B, chan, T, 129] feature maps on the HIP device (the kernels are specialised to 129 frequency bins)"""
    if x.ndim != 4 or x.shape[1] != chan or x.shape[3] != F_BINS:
        raise ValueError(f"{what}: expected a [B, {chan}, T, {F_BINS}] tensor, got {tuple(x.shape)}")
    if not x.is_cuda:
        raise RuntimeError("AVNet's stage modules run on an MI355X HIP device only (no CPU fallback)")


def _check_inference(hip, what):
    if hip.model.training or (torch.is_grad_enabled() and any(p.requires_grad for p in hip.model.parameters())):
        raise NotImplementedError(f"{what}: the stage modules of rtfs_net_amd.AVNet are callable one by one on the inference path only (model.eval() under "
                                  "torch.no_grad()); the training step is one autograd chain behind AVNet.forward")


class StageViews:
    def __init__(self, hip):
        self.hip = hip

    # ---- STFTEncoder.forward (TDAVNet/encoder.py:161-175): [B,L] | [L] | [B,1,L] -> [B,256,T,F] ----
    def encoder(self, x):
        _check_inference(self.hip, "encoder")
        if x.ndim == 1:  # encoder.py:18-25
            x = x.reshape(1, -1)
        elif x.ndim == 3:
            assert x.shape[1] == 1
            x = x.reshape(x.shape[0], -1)
        if not x.is_cuda:
            raise RuntimeError("AVNet's stage modules run on an MI355X HIP device only (no CPU fallback)")
        with torch.no_grad(), torch.cuda.device(x.device):
            pw = self.hip.weights()
            wav = x.to(torch.float32).contiguous()
            B, L = wav.shape
            T = 1 + L // 128
            spec = torch.empty(B * T * F_BINS * 2, device=wav.device)
            lib.call("rtfs_stft_fwd", wav, spec, B, L)
            a_emb = torch.empty(B * T * F_BINS * C, device=wav.device)
            scratch = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device=wav.device)  # (the conv's epilogue accumulates the bottleneck's statistics)
            lib.call("rtfs_enc_conv_fwd", spec, pw.w["enc"], a_emb, scratch, B, T)
            return _nchw(a_emb, B, C, T, F_BINS)

    # ---- audio_bottleneck = ConvNormAct(gLN -> ReLU -> 1x1) (tdavnet.py:59; conv_layers.py:65-129) ----
    def bottleneck(self, a_emb):
        _check_inference(self.hip, "audio_bottleneck")
        _check_map(a_emb, "audio_bottleneck")
        with torch.no_grad(), torch.cuda.device(a_emb.device):
            hip, pw = self.hip, self.hip.weights()
            w = pw.w
            B, _, T, Fq = a_emb.shape
            x = _cl(a_emb)
            stats = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device=x.device)
            lib.call("rtfs_gln_stats", x, stats, B, T * Fq * C)
            a0 = torch.empty_like(x)
            hip._mm("rtfs_bottleneck_fwd", x, stats, w["bn_g"], w["bn_b"], hip._wk(w, "bn_w"), w["bn_bias"], a0, B, T * Fq)
            return _nchw(a0, B, C, T, Fq)

    # ---- RefinementModule.forward (TDAVNet/refinement_module.py:45-62) ----
    def refinement(self, audio, video):
        _check_inference(self.hip, "refinement_module")
        _check_map(audio, "refinement_module")
        if video.ndim != 3 or video.shape[0] != audio.shape[0] or video.shape[1] != 512:
            raise ValueError(f"refinement_module: expected lip features [B, 512, Tv], got {tuple(video.shape)}")
        with torch.no_grad(), torch.cuda.device(audio.device):
            hip, pw = self.hip, self.hip.weights()
            B, _, T, Fq = audio.shape
            T2 = (T - 2) // 2 + 1
            if T2 < 8:
                raise ValueError("input too short: fewer than 8 compressed frames (16 STFT frames) - the 8-tap unfold of the time-path DualPathRNN has no window")
            a0 = _cl(audio)
            R = hip.model.refinement_module.audio_net.repeats
            stats = torch.zeros(1 + 12 * R, B, lib.STAT_STRIDE, dtype=torch.float64, device=a0.device)
            hooked = inner_hooked_modules(hip.model)
            taps = {} if hooked else hip.taps
            all_blocks = hip.tap_all_blocks
            hip.tap_all_blocks = True if hooked else all_blocks
            try:
                s, _ = hip._refine(pw, a0, video, stats, B, T, T2, taps, bottleneck=False)
            finally:
                hip.tap_all_blocks = all_blocks
            if hooked:
                fire_refinement_hooks(hip, pw, taps, stats, a0, audio, video, B, T, T2)
            return _nchw(s, B, C, T, Fq)

    # ---- MaskGenerator.forward + __apply_masks (TDAVNet/mask_generator.py:67-99): -> [B, n_src = 1, 256, T, F] ----
    def mask(self, refined, a_emb):
        _check_inference(self.hip, "mask_generator")
        _check_map(refined, "mask_generator"), _check_map(a_emb, "mask_generator")
        if refined.shape != a_emb.shape:
            raise ValueError(f"mask_generator: refined features {tuple(refined.shape)} and the mixture embedding {tuple(a_emb.shape)} differ in shape")
        with torch.no_grad(), torch.cuda.device(refined.device):
            hip, pw = self.hip, self.hip.weights()
            w = pw.w
            B, _, T, Fq = refined.shape
            s, e = _cl(refined), _cl(a_emb)
            masked = torch.empty_like(s)
            hip._mm("rtfs_mask_fwd", s, w["mask_slope"], hip._wk(w, "mask_w"), w["mask_b"], e, masked, None, B, T * Fq)
            return _nchw(masked, B, C, T, Fq).view(B, 1, C, T, Fq)

    # ---- STFTDecoder.forward (TDAVNet/decoder.py:110-132): [B, 1, 256, T, F], input shape -> [B, 1, L] ----
    def decoder(self, x, input_shape):
        _check_inference(self.hip, "decoder")
        if x.ndim != 5 or x.shape[1] != 1 or x.shape[2] != C or not x.is_cuda:
            raise ValueError(f"decoder: expected the mask generator's [B, 1, {C}, T, {F_BINS}] device tensor, got {tuple(x.shape)}")
        with torch.no_grad(), torch.cuda.device(x.device):
            hip, pw = self.hip, self.hip.weights()
            w = pw.w
            B, L = int(input_shape[0]), int(input_shape[-1])
            T, Fq = x.shape[-2:]
            if T != 1 + L // 128 or Fq != F_BINS:
                raise ValueError(f"decoder: a [{T} x {Fq}] spectrogram does not belong to {L} samples (win 256, hop 128)")
            masked = _cl(x.reshape(B, C, T, Fq))
            tapbuf = torch.empty(B * T * Fq * 32, device=masked.device)
            hip._mm("rtfs_gemm_rows_fwd", masked, hip._wk(w, "dec_w"), None, tapbuf, B * T * Fq, 256, 32)
            frames = torch.empty(B * T * 256, device=masked.device)
            out = torch.empty(B, L, device=masked.device)
            lib.call("rtfs_istft_fwd", tapbuf, frames, out, B, L)
            return out.view(B, 1, L)


# ---- forward hooks ------------------------------------------------------------------------------------------------------------------------------

def _has_hooks(mod):
    return bool(mod._forward_hooks) or bool(mod._forward_pre_hooks)


def _block_children(blk):
    """the sub-modules of one RTFS block whose outputs the stage taps hold (tdanet.py:106-133), in execution order"""
    return [("gateway", blk.gateway), ("projection", blk.projection), ("down0", blk.downsample_layers[0]), ("down1", blk.downsample_layers[1]),
            ("dp_freq", blk.globalatt[0]), ("dp_time", blk.globalatt[1]), ("attn", blk.globalatt[2]), ("globalatt", blk.globalatt),
            ("tfar0", blk.fusion_layers[0]), ("tfar1", blk.fusion_layers[1]), ("concat0", blk.concat_layers[0]), ("residual_conv", blk.residual_conv),
            ("block", blk)]


def stage_modules(model):
    return [model.encoder, model.audio_bottleneck, model.video_bottleneck, model.refinement_module, model.mask_generator, model.decoder]


def inner_served_modules(model):
    """modules inside the refinement module whose hooks fire_refinement_hooks serves"""
    rm = model.refinement_module
    mods = []
    for i in range(rm.audio_net.repeats if not rm.audio_net.shared else 1):
        mods += [m for _, m in _block_children(rm.audio_net.get_block(i))]
    fusion = rm.crossmodal_fusion.get_fusion_block(0)
    return mods + [rm.video_net.get_block(0), fusion, fusion.audio_lstm]


def inner_hooked_modules(model):
    return [m for m in inner_served_modules(model) if _has_hooks(m)]


def any_served_hooks(model):
    """cheap per-forward check (a few dozen modules): does any module this file serves carry a forward (pre-)hook?"""
    served = model.__dict__.get("_served_hook_modules")
    if served is None:
        served = model.__dict__["_served_hook_modules"] = stage_modules(model) + inner_served_modules(model)
    for m in served:
        if m._forward_hooks or m._forward_pre_hooks:
            return True
    return False


def refuse_unserved_hooks(model):
    """a hook on a module whose output the HIP path never forms (a Conv2d inside a ConvNormAct, a norm layer, an SRU cell) cannot fire: say so"""
    served = set(map(id, stage_modules(model) + inner_served_modules(model)))
    video = model.refinement_module.video_net
    video_ids = set(map(id, video.modules()))  # (the VP block's torch sub-modules fire their own hooks when the block runs as PyTorch glue)
    for name, m in model.named_modules():
        if m is model or id(m) in served or id(m) in video_ids:
            continue
        if _has_hooks(m):
            raise NotImplementedError(f"rtfs_net_amd: a forward hook on '{name}' cannot be served - the HIP path fuses that module into its neighbours and never "
                                      "forms its output; hook one of the stage modules, an RTFS block or its direct children, the VP block or the CAF cell")


def _fire(mod, inputs, output):
    for hook in list(mod._forward_pre_hooks.values()):
        if hook(mod, inputs) is not None:
            raise NotImplementedError("rtfs_net_amd: forward pre-hooks may observe, not replace, a fused module's inputs")
    for hook in list(mod._forward_hooks.values()):
        if hook(mod, inputs, output) is not None:
            raise NotImplementedError("rtfs_net_amd: forward hooks may observe, not replace, a fused module's output (the kernels downstream have already run)")


def fire_refinement_hooks(hip, pw, taps, stats, a0_cl, audio_nchw, video, B, T, T2):
    """forward hooks of the modules inside the refinement module, in the reference's execution order (block 0 and its children, VP block, CAF cell,
    blocks 1..R-1 and theirs), each with (inputs, output) in the reference's NCHW layout - materialised only for modules that carry a hook"""
    model = hip.model
    rm = model.refinement_module
    R = rm.audio_net.repeats
    TF, lo = T * F_BINS, T2 * F2
    blocks = pw.blocks
    full = lambda t, ch=H: _nchw(t, B, ch, T, F_BINS)  # noqa: E731
    low = lambda t: _nchw(t, B, H, T2, F2)  # noqa: E731

    def minus(x, y):  # x - y on channels-last buffers (rtfs_axpy on a copy)
        out = x.clone()
        lib.call("rtfs_axpy", y, -1.0, out, out.numel())
        return out

    def plus(x, y):
        out = x.clone()
        lib.call("rtfs_axpy", y, 1.0, out, out.numel())
        return out

    def norm_act(x, slot, gamma, beta, act, slope, rows):
        y = torch.empty_like(x)
        lib.call("rtfs_norm_act_fwd", x, slot, gamma, beta, act, float(slope), y, B, rows, H)
        return y

    last_only = R == 1
    caf_cl = None
    for i in range(R):
        blk = rm.audio_net.get_block(i)
        bw = blocks[0] if len(blocks) == 1 else blocks[i]
        sfx = "" if i == 0 else f"#{i}"
        st = stats[1 + 12 * i: 13 + 12 * i]
        hooked = {name for name, m in _block_children(blk) if _has_hooks(m)}
        if hooked:
            # block input (refinement_module.py:52,60): a0 | CAF output + a0 | previous block output + a0 - what the residual kernels wrote
            s_in = a0_cl if i == 0 else (taps["caf_plus_a0"] if i == 1 else taps[f"block#{i - 1}"])
            out_cl = taps["block0"] if i == 0 else (taps["block" + sfx] if i == R - 1 else minus(taps["block" + sfx], a0_cl))
            view = {}

            def get(name):
                if name in view:
                    return view[name]
                if name == "s_in":
                    v = full(s_in, C)
                elif name == "gateway_cl":
                    v = torch.empty_like(s_in)
                    lib.call("rtfs_gateway_fwd", s_in, bw["gw"], bw["gb"], float(bw["gslope"]), v, B * TF)
                elif name == "gateway":
                    v = full(get("gateway_cl"), C)
                elif name == "projection":
                    v = full(norm_act(taps["y0" + sfx], st[0], bw["pg"], bw["pbe"], 1, bw["pslope"], TF))
                elif name == "down0_cl":
                    v = norm_act(taps["D0" + sfx], st[1], bw["d0"][2], bw["d0"][3], 0, 0.0, TF)
                elif name == "down0":
                    v = full(get("down0_cl"))
                elif name == "down1":
                    v = low(norm_act(taps["D1" + sfx], st[2], bw["d1"][2], bw["d1"][3], 0, 0.0, lo))
                elif name in ("pooled", "dp_freq", "dp_time", "attn", "tfar1"):
                    v = low(taps[name + sfx])
                elif name == "tfar0":
                    v = full(taps["tfar0" + sfx])
                elif name == "concat0_cl":  # InjectionMultiSum of concat_layers[0]: n(cl) * sigmoid(n(cgate))^ + n(cg)^ (fusion.py:54-69)
                    cl_, cg_, cgate_ = (bw[f"concat_layers.0.{e}"] for e in ("local_embedding", "global_embedding", "global_gate"))
                    v = torch.empty_like(taps["cl" + sfx])
                    lib.call("rtfs_tfar_mix_fwd", taps["cl" + sfx], st[9], cl_[2], cl_[3], taps["cgate" + sfx], st[11], cgate_[2], cgate_[3], taps["cg" + sfx], st[10],
                             cg_[2], cg_[3], v, B, T, F_BINS, T2, F2)
                elif name == "concat0":
                    v = full(get("concat0_cl"))
                elif name == "expanded":  # concat output + downsample_layers[0]'s output: the residual conv's input (tdanet.py:129-131)
                    v = full(plus(get("concat0_cl"), get("down0_cl")))
                elif name == "residual_conv":  # block output = residual_conv(expanded) + gateway output
                    v = full(minus(out_cl, get("gateway_cl")), C)
                elif name == "block":
                    v = full(out_cl, C)
                else:
                    raise KeyError(name)
                view[name] = v
                return v

            inputs = {"gateway": ("s_in",), "projection": ("gateway",), "down0": ("projection",), "down1": ("down0",), "dp_freq": ("pooled",),
                      "dp_time": ("dp_freq",), "attn": ("dp_time",), "globalatt": ("pooled",), "tfar0": ("down0", "attn"), "tfar1": ("down1", "attn"),
                      "concat0": ("tfar0", "tfar1"), "residual_conv": ("expanded",), "block": ("s_in",)}
            for name, mod in _block_children(blk):
                if name in hooked:
                    _fire(mod, tuple(get(k) for k in inputs[name]), get("attn" if name == "globalatt" else name))
        if i == 0:
            vb = rm.video_net.get_block(0)
            fusion = rm.crossmodal_fusion.get_fusion_block(0)
            need_caf = _has_hooks(fusion) or _has_hooks(fusion.audio_lstm)
            if _has_hooks(vb) and not hip.vp_ran_as_modules:  # (as PyTorch modules - RTFS_DISABLE=vp_hip - torch has fired them already)
                _fire(vb, (video,), taps["vp"])
            if need_caf:
                caf_cl = taps["caf"] if last_only else minus(taps["caf_plus_a0"], a0_cl)
                block0, caf = full(taps["block0"], C), full(caf_cl, C)
                if _has_hooks(fusion.audio_lstm):  # ATTNFusionCell.forward(audio, video) -> fused audio (layers/fusion.py:252-274)
                    _fire(fusion.audio_lstm, (block0, taps["vp"]), caf)
                if _has_hooks(fusion):  # ATTNFusion.forward -> (audio, video) (TDAVNet/fusion.py:204-212)
                    _fire(fusion, (block0, taps["vp"]), (caf, taps["vp"]))
