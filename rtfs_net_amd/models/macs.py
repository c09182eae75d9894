"""Analytic multiply-accumulate report for a 2 s input, in the layout of BaseAVModel.get_MACs
(/root/reference/src/models/TDAVNet/base_av_model.py:61-118).  Counts convolutions, linear maps and attention
products (what thop counts); SURVEY.md §6 cross-checks these formulas against the published 21.9/30.5/56.4 G."""
from __future__ import annotations


def _params(mod):
    return int(sum(p.numel() for p in mod.parameters() if p.requires_grad) / 1000)


def audio_block_macs(T=251, F=129, C=256, H=64):
    T2, F2 = (T - 2) // 2 + 1, (F - 2) // 2 + 1
    TF, lo = T * F, T2 * F2
    m = C * TF  # gateway
    m += C * H * TF  # projection
    m += 16 * H * TF + 16 * H * lo  # downsample
    for npos, nseq in ((F2, T2), (T2, F2)):  # dual path
        L = npos - 7
        m += nseq * L * (512 * 256 + 3 * 64 * 192) + nseq * L * 64 * 64 * 8
    m += lo * H * 96 + 4 * (T2 * T2 * 256 + T2 * T2 * 1024) + lo * H * H  # attention
    m += 16 * H * (TF + lo) + 4 * 16 * H * lo + 16 * H * TF + 2 * 16 * H * lo  # TFAR depth-wise convs
    m += H * C * TF  # residual conv
    return m


def macs_report(model, seconds=2):
    T, F, C = 1 + seconds * 16000 // 128, 129, 256
    TF = T * F
    R = model.refinement_module.audio_net.repeats
    enc = 18 * C * TF
    bn = C * C * TF
    blk = audio_block_macs(T, F)
    caf = 2 * C * TF + 50 * (1024 * 2 + 256 * 2)
    mask = C * C * TF
    dec = 18 * C * TF
    vid = 10_000_000
    rm = R * blk + vid + caf
    rows = [enc, _params(model.encoder), bn, _params(model.audio_bottleneck), 0, _params(model.video_bottleneck), rm,
            _params(model.refinement_module), R * blk, _params(model.refinement_module.audio_net), vid,
            _params(model.refinement_module.video_net), caf, _params(model.refinement_module.crossmodal_fusion), mask,
            _params(model.mask_generator), dec, _params(model.decoder), enc + bn + rm + mask + dec, _params(model)]
    rows = ["{:,}".format(int(r / 1e6) if i % 2 == 0 else r) for i, r in enumerate(rows)]
    return (
        "RTFS-Net (rtfs_net_amd)\n"
        "Encoder ------------- MACs: {:>8} M    Params: {:>6} K\n"
        "Audio BN ------------ MACs: {:>8} M    Params: {:>6} K\n"
        "Video BN ------------ MACs: {:>8} M    Params: {:>6} K\n"
        "RefinementModule ---- MACs: {:>8} M    Params: {:>6} K\n"
        "   AudioNet --------- MACs: {:>8} M    Params: {:>6} K\n"
        "   VideoNet --------- MACs: {:>8} M    Params: {:>6} K\n"
        "   FusionNet -------- MACs: {:>8} M    Params: {:>6} K\n"
        "Mask Generator ------ MACs: {:>8} M    Params: {:>6} K\n"
        "Decoder ------------- MACs: {:>8} M    Params: {:>6} K\n"
        "Total --------------- MACs: {:>8} M    Params: {:>6} K\n"
    ).format(*rows)
