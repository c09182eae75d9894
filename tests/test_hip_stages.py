"""GPU parity, stage by stage: every HIP stage boundary of one RTFS-Net forward against the oracle
(oracle/avnet_ref.py, itself pinned to the reference by tests/golden) on the same seeded inputs/weights.

Tolerances are relative L2 per stage boundary: 2e-5 for single kernels fed exact inputs would be
fp32 round-off; here errors accumulate along the chain, so stage taps use 2e-4 and the waveform uses
the north-star bound 1e-3 (BASELINE.json).
"""
import pytest
import torch
import torch.nn.functional as F

from util import cl_to_nchw, make_model, rel, synth

pytestmark = pytest.mark.gpu

STAGE_TOL = 2e-4
WAVE_TOL = 1e-3


@pytest.fixture(scope="module")
def run():
    """One forward of RTFS-Net-2 (same block weights as -4/-6/-12) at B=2, L=16000 on HIP and in the oracle, with taps."""
    from oracle.avnet_ref import avnet_forward

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    B, L, Tv, R = 2, 16000, 25, 2
    model, sd, cfg = make_model(R, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    model._hip.taps = {}
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
    torch.cuda.synchronize()
    taps = {k: v.detach().float().cpu() for k, v in model._hip.taps.items()}
    model._hip.taps = None
    otaps = {}
    with torch.no_grad():
        ref = avnet_forward(sd, cfg, mix, emb, taps=otaps)
    T = 1 + L // 128
    T2 = (T - 2) // 2 + 1
    return dict(B=B, L=L, T=T, T2=T2, Tv=Tv, sd=sd, out=out.cpu(), ref=ref, taps=taps, otaps=otaps, mix=mix)


def _full(r, name, C):
    return cl_to_nchw(r["taps"][name], r["B"], r["T"], 129, C)


def _low(r, name):
    return cl_to_nchw(r["taps"][name], r["B"], r["T2"], 64, 64)


def test_stft(run):
    from oracle.avnet_ref import stft_frames

    spec = cl_to_nchw(run["taps"]["spec"], run["B"], run["T"], 129, 2)
    assert rel(spec, stft_frames(run["mix"], 256, 128)) < 2e-6


def test_encoder_conv(run):
    assert rel(_full(run, "a_emb", 256), run["otaps"]["a_emb"]) < 1e-5


def test_bottleneck(run):
    assert rel(_full(run, "a0", 256), run["otaps"]["a0"]) < 2e-5


def test_projection(run):
    sd, p = run["sd"], "refinement_module.audio_net.blocks.projection.full_layer."
    y0 = _full(run, "y0", 64)
    got = F.prelu(F.group_norm(y0, 1, sd[p + "3.norm.weight"], sd[p + "3.norm.bias"], 1e-5), sd[p + "4.weight"])
    assert rel(got, run["otaps"]["block0.proj"]) < 5e-5


def test_downsample(run):
    sd, p = run["sd"], "refinement_module.audio_net.blocks.downsample_layers."
    d0 = F.group_norm(_full(run, "D0", 64), 1, sd[p + "0.full_layer.3.norm.weight"], sd[p + "0.full_layer.3.norm.bias"], 1e-5)
    d1 = F.group_norm(_low(run, "D1"), 1, sd[p + "1.full_layer.3.norm.weight"], sd[p + "1.full_layer.3.norm.bias"], 1e-5)
    assert rel(d0, run["otaps"]["block0.ds0"]) < 5e-5
    assert rel(d1, run["otaps"]["block0.ds1"]) < 5e-5


def test_pool(run):
    assert rel(_low(run, "pooled"), run["otaps"]["block0.pooled"]) < 5e-5


def test_dual_path_freq(run):
    assert rel(_low(run, "dp_freq"), run["otaps"]["block0.globalatt.0"]) < STAGE_TOL


def test_dual_path_time(run):
    assert rel(_low(run, "dp_time"), run["otaps"]["block0.globalatt.1"]) < STAGE_TOL


def test_attention(run):
    assert rel(_low(run, "attn"), run["otaps"]["block0.globalatt.2"]) < STAGE_TOL


def test_tfar(run):
    assert rel(_full(run, "tfar0", 64), run["otaps"]["block0.fused0"]) < STAGE_TOL
    assert rel(_low(run, "tfar1"), run["otaps"]["block0.fused1"]) < STAGE_TOL


def test_block0(run):
    assert rel(_full(run, "block0", 256), run["otaps"]["block0"]) < STAGE_TOL


def test_vp_block(run):
    """a9 / f3: the one-launch VP block (csrc/vp.hip) against the oracle's TDANetBlock(is2d=False) + GlobalAttention"""
    assert rel(run["taps"]["vp"], run["otaps"]["vp"]) < 5e-5


@pytest.mark.parametrize("Tv", [1, 2, 3, 4, 5, 7, 8, 9, 25, 50, 63, 64, 65, 100])
def test_vp_block_lengths_match_glue(Tv):
    """every supported length (ragged down-sampling chains, two-pass lane = t loops) against the PyTorch-glue modules"""
    import copy

    import torch

    from rtfs_net_amd import lib
    from util import make_model

    model, _, _ = make_model(2, "cuda")
    pw = model._hip.weights().w
    x = torch.randn(3, 512, Tv, generator=torch.Generator().manual_seed(Tv)).cuda()
    out = torch.empty_like(x)
    lib.call("rtfs_vp_block_fwd", x, pw["vp"], pw["vp_pe"], out, 3, Tv)
    with torch.no_grad():
        ref = model.refinement_module.video_net.get_block(0)(x)
    assert rel(out.cpu(), ref.cpu()) < 2e-5


def test_caf(run):
    got = _full(run, "caf_plus_a0", 256) - _full(run, "a0", 256)
    assert rel(got, run["otaps"]["caf"]) < STAGE_TOL


def test_mask(run):
    got = _full(run, "masked", 256)
    assert rel(got, run["otaps"]["masked"][:, 0]) < 5e-4


def test_waveform(run):
    assert run["out"].shape == run["ref"].shape == (run["B"], 1, run["L"])
    assert rel(run["out"], run["ref"]) < WAVE_TOL


# ---- isolated kernels on their own random inputs ------------------------------------------------
def test_sru_scan_isolated():
    """rtfs_sru_scan_fwd against oracle/sru_ref.py on random U (both layer kinds, ragged L)."""
    from oracle.sru_ref import sru_cell_forward
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(1)
    S, L, d = 37, 29, 32
    x = torch.randn(L, S, 64, generator=g)
    wc, bias = torch.rand(128, generator=g) * 2 - 1, torch.randn(128, generator=g) * 0.3
    for k in (4, 3):
        W = torch.randn(64, 64 * k, generator=g) * 0.3
        href, _ = sru_cell_forward(x, W, wc, bias, torch.ones(1), d)
        U = (x.reshape(L * S, 64) @ W).view(L, S, 64, k)  # [l][s][lane][m]
        if k == 4:
            Ud = U.permute(1, 0, 2, 3).contiguous()  # [s][l][lane][4]
        else:
            Ud = U.permute(1, 0, 3, 2).contiguous()  # [s][l][m][lane]
        Xd = x.permute(1, 0, 2).contiguous().cuda()
        H = torch.empty(S * L * 64, device="cuda")
        lib.call("rtfs_sru_scan_fwd", Ud.cuda(), Xd, wc.cuda(), bias.cuda(), 1.0, H, S, L, k)
        got = H.view(S, L, 64).permute(1, 0, 2).cpu()
        assert rel(got, href) < 2e-5, k


def test_istft_roundtrip():
    """decoder taps of an identity-like spectrum: STFT -> (taps carrying the spectrum in the centre tap) -> iSTFT == input."""
    from rtfs_net_amd import lib

    B, L = 3, 4096 + 128 * 5
    T = 1 + L // 128
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, L, generator=g).cuda()
    spec = torch.empty(B * T * 129 * 2, device="cuda")
    lib.call("rtfs_stft_fwd", x, spec, B, L)
    taps = torch.zeros(B, T, 129, 32, device="cuda")
    sp = spec.view(B, T, 129, 2)
    taps[..., 4] = sp[..., 0]       # o=0, kt=1, kf=1 -> same pixel
    taps[..., 9 + 4] = sp[..., 1]   # o=1
    frames = torch.empty(B * T * 256, device="cuda")
    out = torch.empty(B, L, device="cuda")
    lib.call("rtfs_istft_fwd", taps, frames, out, B, L)
    assert rel(out, x) < 5e-6


@pytest.mark.parametrize("S,L", [(5, 57), (3, 118), (2, 9), (4, 32), (1, 33), (7, 1), (600, 40), (2100, 9)])
def test_sru_layer_fused_matches_gemm_plus_scan(S, L):
    """rtfs_sru_layer_fwd (projection on MFMA inside the recurrence) == rtfs_gemm_rows_fwd(64->192) + rtfs_sru_scan_fwd, ragged lengths
    (chunk boundaries at 32, single-step sequences), with and without the training saves (cell states, pre-activations).  The kernel forms - one
    workgroup per (sequence, direction) up to 256 sequences, one wave per (sequence, direction) below 2048 (inference), 4-wave workgroups below 2048,
    8-wave workgroups above (training / large batches) - give the same bits, each also named explicitly through rtfs_sru_layer_fwd_form."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(S * 131 + L)
    h = torch.randn(S * L * 64, generator=g).cuda()
    W = (torch.randn(192, 64, generator=g) * 0.2).cuda()
    wc, bias = (torch.randn(128, generator=g) * 0.5).cuda(), (torch.randn(128, generator=g) * 0.5).cuda()
    U = torch.empty(S * L * 192, device="cuda")
    lib.call("rtfs_gemm_rows_fwd", h, W, None, U, S * L, 64, 192)
    ref = torch.empty_like(h)
    lib.call("rtfs_sru_scan_fwd", U, h, wc, bias, 0.7, ref, S, L, 3)
    out, out2, cst, U2 = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h), torch.empty_like(U)
    lib.call("rtfs_sru_layer_fwd", h, W, wc, bias, 0.7, out, None, None, S, L)
    lib.call("rtfs_sru_layer_fwd", h, W, wc, bias, 0.7, out2, cst, U2, S, L)
    assert float((out - ref).abs().max()) < 2e-5 and float((out2 - ref).abs().max()) < 2e-5
    assert torch.equal(out, out2)  # (the inference form of this S and the training form: same products in the same order, same recurrence)
    for form in (1, 2, 3):  # one wave per sequence / per (sequence, direction) / one workgroup per (sequence, direction) with the gates split over its waves
        outf = torch.full_like(h, float("nan"))
        lib.call("rtfs_sru_layer_fwd_form", h, W, wc, bias, 0.7, outf, None, None, S, L, form)
        assert torch.equal(outf, out), form
    assert float((U2 - U).abs().max()) < 2e-5
    with pytest.raises(RuntimeError):  # saves come as a pair
        lib.call("rtfs_sru_layer_fwd", h, W, wc, bias, 0.7, out2, cst, None, S, L)


@pytest.mark.parametrize("S,L", [(37, 57), (2100, 57), (130, 118), (9, 33)])
def test_sru_layer_six_term_split_matches_fp32(S, L):
    """rtfs_sru_layer_fwd_bf16 with terms = 6 (the 64 -> 192 projection as six bf16 MFMAs per product, weight rows split once into three planes in
    LDS, h_prev fragments split in registers; recurrence in fp32) against the fp32 entry point: h, and in the training form the cell states and
    pre-activations, to round-off; both workgroup sizes (8 waves from 2048 sequences), ragged last chunks."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(S + L)
    h = torch.randn(S, L, 64, generator=g).cuda()
    W = (torch.randn(192, 64, generator=g) * 0.2).cuda()
    wc, bias = (torch.randn(128, generator=g) * 0.5).cuda(), (torch.randn(128, generator=g) * 0.5).cuda()
    ref, cref, uref = torch.empty_like(h), torch.empty_like(h), torch.empty(S * L * 192, device="cuda")
    lib.call("rtfs_sru_layer_fwd", h, W, wc, bias, 0.7, ref, cref, uref, S, L)
    out, out2, cst, U2 = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h), torch.empty_like(uref)
    lib.call("rtfs_sru_layer_fwd_bf16", h, W, wc, bias, 0.7, out, None, None, S, L, 6)
    lib.call("rtfs_sru_layer_fwd_bf16", h, W, wc, bias, 0.7, out2, cst, U2, S, L, 6)
    assert torch.equal(out, out2)
    # (two fp32-accurate projections that round differently, carried through the recurrence: the relative L2 bound is the check; the largest single element
    # among 7.7M - a cell state of magnitude ~5 after a run of open forget gates - was 5.3e-5 off at S = 2100)
    assert float((out - ref).abs().max()) < 2e-4 and float((cst - cref).abs().max()) < 2e-4 and float((U2 - uref).abs().max()) < 2e-5
    assert rel(out, ref) < 2e-6 and rel(U2, uref) < 2e-6


@pytest.mark.parametrize("B,T2", [(5, 125), (13, 40), (8, 40), (5, 77), (9, 16), (10, 125), (17, 70), (23, 50), (5, 250)])
@pytest.mark.parametrize("dim", [4, 3])
def test_unfold_gemm_entry_flattened_tiles(B, T2, dim):
    """rtfs_dp_unfold_gemm_fwd in isolation (LN4D over channels + 8-tap unfold + layer-0 GEMM, rnn_layers.py:146-150) against float64 on
    the CPU, at sizes that take the large-batch kernel (tiles cut from the flattened row index: 2- and 3-sequence tiles, ragged end)
    a few that take the small-batch one, and three that take the weight-stationary kernels: variant 0 = the fast-FIR form (>= 512 tiles of 63 virtual
    rows; pair rows per sequence Lv = 29 / 60 / 32 / 22: two-, three- and four-sequence tiles, odd and even window counts; three half-rate 4-tap
    correlations, so its sums are ordered differently: agreement to round-off, not bit for bit), variant 3 = the direct form (>= 1024 flattened 64-row
    tiles; its LayerNorm uses v_rsq_f32, so it agrees with the LDS-staged kernels to 1 ulp of rstd, not bit for bit)."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(100 * B + T2 + dim)
    G = torch.randn(B, T2, 64, 64, generator=g)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    Wt = torch.randn(256, 512, generator=g) * 0.05
    x = G.double()
    xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    seqs = xn.reshape(B * T2, 64, 64) if dim == 4 else xn.permute(0, 2, 1, 3).reshape(B * 64, T2, 64)  # [S][npos][64]
    L = seqs.shape[1] - 7
    win = torch.stack([seqs[:, k:k + L] for k in range(8)], dim=2).reshape(seqs.shape[0], L, 512)  # k index = tap * 64 + channel
    want = win @ Wt.double().t()
    U = torch.full((seqs.shape[0] * L * 256,), float("nan"), device="cuda")
    lib.call("rtfs_dp_unfold_gemm_fwd", G.cuda(), gamma.cuda(), beta.cuda(), Wt.cuda(), U, B, T2, dim, 0)
    assert rel(U.view(want.shape), want) < 2e-6
    U1 = torch.full_like(U, float("nan"))
    lib.call("rtfs_dp_unfold_gemm_fwd", G.cuda(), gamma.cuda(), beta.cuda(), Wt.cuda(), U1, B, T2, dim, 1)  # per-sequence tiles
    U2 = torch.full_like(U, float("nan"))
    lib.call("rtfs_dp_unfold_gemm_fwd", G.cuda(), gamma.cuda(), beta.cuda(), Wt.cuda(), U2, B, T2, dim, 2)  # LDS-staged flattened tiles: the same bits
    assert torch.equal(U2, U1)
    weight_stationary = (seqs.shape[0] * L + 63) // 64 >= 1024 and L >= 32
    fast_fir = (seqs.shape[0] * ((L + 2) // 2) + 62) // 63 >= 512 and (L + 2) // 2 >= 21
    assert rel(U, U2) < 1e-6 and (weight_stationary or fast_fir or torch.equal(U, U2))
    U3 = torch.full_like(U, float("nan"))
    lib.call("rtfs_dp_unfold_gemm_fwd", G.cuda(), gamma.cuda(), beta.cuda(), Wt.cuda(), U3, B, T2, dim, 3)  # direct weight-stationary form where eligible
    assert rel(U3.view(want.shape), want) < 2e-6 and rel(U3, U2) < 1e-6 and (weight_stationary or torch.equal(U3, U2))
    if fast_fir:  # every output ROW on its own (a mis-addressed halo unit or pair row shows in single rows long before it shows in the norm)
        rows = (U.view(want.shape).double().cpu() - want).norm(dim=-1) / want.norm(dim=-1)
        assert float(rows.max()) < 5e-6, (float(rows.max()), int(rows.argmax()))
    with pytest.raises(RuntimeError):
        lib.call("rtfs_dp_unfold_gemm_fwd", G.cuda(), gamma.cuda(), beta.cuda(), Wt.cuda(), U1, B, T2, dim, 7)


@pytest.mark.parametrize("B,T2", [(5, 125), (13, 40), (17, 70), (23, 50), (5, 250), (10, 125)])
@pytest.mark.parametrize("dim", [4, 3])
def test_unfold_gemm_entry_six_term_split(B, T2, dim):
    """rtfs_dp_unfold_gemm_fwd_bf16 with terms = 6 (fp32 operands split into three bfloat16 values, six MFMAs per product) in isolation against
    float64 on the CPU.  From 512 flattened 64-row tiles on (and windows per sequence L >= 32) variant 0 is the weight-stationary kernel of this
    split (unfold_ws6_kernel: operands split once, two tap halves of a column block added through LDS; tiles over two and three sequences, ragged
    ends, tile ranges with an odd count = the dummy tile); below that and as variant 2, the LDS-staged kernel that splits in registers.  The split
    keeps all 24 mantissa bits of both operands: same bound as the fp32 kernels, every output row on its own as well."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(100 * B + T2 + dim + 6)
    G = torch.randn(B, T2, 64, 64, generator=g)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    Wt = torch.randn(256, 512, generator=g) * 0.05
    x = G.double()
    xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    seqs = xn.reshape(B * T2, 64, 64) if dim == 4 else xn.permute(0, 2, 1, 3).reshape(B * 64, T2, 64)  # [S][npos][64]
    L = seqs.shape[1] - 7
    win = torch.stack([seqs[:, k:k + L] for k in range(8)], dim=2).reshape(seqs.shape[0], L, 512)  # k index = tap * 64 + channel
    want = win @ Wt.double().t()
    out = {}
    for variant in (0, 2):
        U = torch.full((seqs.shape[0] * L * 256,), float("nan"), device="cuda")
        lib.call("rtfs_dp_unfold_gemm_fwd_bf16", G.cuda(), gamma.cuda(), beta.cuda(), Wt.cuda(), U, B, T2, dim, variant, 6)
        out[variant] = U
        assert rel(U.view(want.shape), want) < 1e-6
        rows = (U.view(want.shape).double().cpu() - want).norm(dim=-1) / want.norm(dim=-1)
        assert float(rows.max()) < 3e-6, (variant, float(rows.max()), int(rows.argmax()))
    weight_stationary = (seqs.shape[0] * L + 63) // 64 >= 512 and L >= 32
    assert rel(out[0], out[2]) < 1e-6 and (weight_stationary != torch.equal(out[0], out[2]))


def test_resid_proj_fusion_matches_separate_calls():
    """Blocks 1..R-2 compute the next block's projection inside the residual kernel: same waveform as the separate kernels (the only
    difference is the summation order of the 256-term projection dot products)."""
    import os

    model, sd, cfg = make_model(4, "cuda")
    mix, _, emb = synth.synth_inputs(3, 16000, 25)
    with torch.no_grad():
        fused = model(mix.cuda(), emb.cuda())
        model._hip.fuse["proj"] = False
        plain = model(mix.cuda(), emb.cuda())
    assert rel(fused, plain) < 2e-6 and not torch.equal(fused, plain)


@pytest.mark.parametrize("R", [1, 2, 4])
def test_caf_fusion_matches_separate_calls(R):
    """Block 0's residual kernel applies the CAF cell's audio side in its epilogue (rtfs_resid_caf_fwd; the block output never reaches HBM,
    the residual stream doubles as the cell's a0 stream).  Against the separate rtfs_resid_fwd + rtfs_caf_fuse_fwd (+ rtfs_proj_fwd) calls:
    the same arithmetic in the same order when block 1's projection is not fused (R = 1 has no a0 and no block 1), the projection's
    summation order otherwise."""
    import os

    model, sd, cfg = make_model(R, "cuda")
    mix, _, emb = synth.synth_inputs(3, 16000, 25)

    def run(proj=True, caf=True):
        model._hip.fuse.update(proj=proj, caf=caf)
        with torch.no_grad():
            return model(mix.cuda(), emb.cuda())

    fused, plain = run(), run(caf=False)
    assert rel(fused, plain) < 2e-6
    fused_np, plain_np = run(proj=False), run(proj=False, caf=False)
    assert rel(fused_np, plain_np) < 2e-7
    if R > 1:
        assert not torch.equal(fused, fused_np)  # the fused projection did run


def test_resid_caf_entry_rejects_bad_arguments():
    """rtfs_resid_caf_fwd: a fused projection needs add_input; Tv must not exceed T."""
    from rtfs_net_amd import lib

    z = torch.zeros(64 * 129 * 256, device="cuda")
    st = torch.zeros(16, dtype=torch.float64, device="cuda")
    v = torch.zeros(256, device="cuda")
    args = lambda Tv, add, wp: (z, st, v, v, z, st, v, v, z, st, v, v, z, st, v, v, z, v, z, v, v, 0.25, v, v, v, v, z, z, Tv, add, z, wp, v, z, st,  # noqa: E731
                                1, 16, 8, 0)
    with pytest.raises(RuntimeError):
        lib.call("rtfs_resid_caf_fwd", *args(4, 0, z))
    with pytest.raises(RuntimeError):
        lib.call("rtfs_resid_caf_fwd", *args(17, 1, None))
    with pytest.raises(RuntimeError):  # unknown kernel form
        lib.call("rtfs_resid_caf_fwd", *(args(4, 1, z)[:-1] + (9,)))


@pytest.mark.parametrize("B,L", [(3, 16000), (2, 32000), (2, 12100), (1, 2048)])
def test_trio_fusion_matches_separate_calls(B, L):
    """rtfs_dwconv_trio_fwd + rtfs_pool_add_fwd (one pass over D0 for D1's stride-2 convolution, the pooling and fusion_layers[0]'s local
    embedding) against the three separate kernels: even and odd frame counts (2- and 3-row pooling windows), ragged tiles, end to end."""
    import os

    model, sd, cfg = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, max(8, L // 640))
    with torch.no_grad():
        fused = model(mix.cuda(), emb.cuda())
        model._hip.fuse["trio"] = False
        plain = model(mix.cuda(), emb.cuda())
    assert rel(fused, plain) < 2e-6


@pytest.mark.parametrize("B,T", [(2, 251), (3, 126), (1, 17), (2, 95)])
def test_trio_entry_against_separate_entries(B, T):
    """The entry points in isolation: every output tensor and both statistics slots."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(7 * B + T)
    T2 = (T - 2) // 2 + 1
    dev = "cuda"
    D0 = torch.randn(B, T, 129, 64, generator=g).to(dev)
    gam, bet = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
    g1, b1 = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
    w1, w2, bias2 = (torch.randn(16, 64, generator=g) * 0.2).to(dev), (torch.randn(16, 64, generator=g) * 0.2).to(dev), torch.randn(64, generator=g).to(dev)
    st0 = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device=dev)
    st0[:, 0] = D0.double().sum((1, 2, 3))
    st0[:, 1] = (D0.double() ** 2).sum((1, 2, 3))
    full = lambda: torch.full((B * T * 129 * 64,), float("nan"), device=dev)  # noqa: E731
    low = lambda: torch.full((B * T2 * 64 * 64,), float("nan"), device=dev)  # noqa: E731
    sts = lambda: torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device=dev)  # noqa: E731
    l0a, D1a, Ga, s1a, s2a = full(), low(), low(), sts(), sts()
    lib.call("rtfs_dwconv_fwd", D0, st0, gam, bet, 0.0, 1, 1, 1, [w1], [None], [l0a], [s1a], B, T, 129)
    lib.call("rtfs_dwconv_fwd", D0, st0, gam, bet, 0.0, 1, 2, 1, [w2], [bias2], [D1a], [s2a], B, T, 129)
    lib.call("rtfs_pool_fwd", D0, st0, gam, bet, D1a, s2a, g1, b1, Ga, B, T, T2)
    l0b, D1b, P, Gb, s1b, s2b = full(), low(), low(), low(), sts(), sts()
    lib.call("rtfs_dwconv_trio_fwd", D0, st0, gam, bet, w1, l0b, s1b, w2, bias2, D1b, s2b, P, B, T, T2)
    lib.call("rtfs_pool_add_fwd", P, D1b, s2b, g1, b1, Gb, B, T2)
    assert torch.equal(l0a, l0b) and torch.equal(D1a, D1b)
    assert rel(Gb, Ga) < 1e-6 and not torch.isnan(Gb).any()
    assert rel(s1b[:, :2], s1a[:, :2]) < 1e-6 and rel(s2b[:, :2], s2a[:, :2]) < 1e-6
    with pytest.raises(RuntimeError):
        lib.call("rtfs_dwconv_trio_fwd", D0, st0, gam, bet, w1, l0b, s1b, w2, bias2, D1b, s2b, P, B, T, T2 + 1)


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_resid_large_batch_forms_match(dtype):
    """At large batch the projection-carrying residual kernels (blocks 1..R-2 and block 0 with the CAF cell) run as ONE workgroup per CU:
    variant 3 = eight waves, four that only issue MFMAs and four that own every global access and the element-wise work (gemm.hip
    resid_ws_kernel, the default), variant 2 = four waves with the whole register file (resid_kernel DEEP), variant 1 = the two-workgroup
    form small batches take.  Per element the same arithmetic in the same order; the workgroups own different numbers of tiles, so the fp32
    partial sums behind the projection's gLN statistics group differently (1e-7 level)."""
    model, sd, cfg = make_model(3, "cuda")
    model.set_compute_dtype(dtype)
    mix, _, emb = synth.synth_inputs(10, 16000 + 2048, 25)  # 287 tiles (the last one 14 pixels: an empty second half) x 10 utterances >= 2048
    outs = {}
    with torch.no_grad():
        for v in (3, 2, 1, 0):
            model._hip.variants["resid"] = v
            outs[v] = model(mix.cuda(), emb.cuda())
    model._hip.variants["resid"] = 0
    # the library's choice at this size is variant 3.  Inside the whole suite the bf16x3 case used to differ between the two runs about once in
    # 15 (never alone): the video-branch kernels on the side stream carried packed-fp32 op_sel instructions, which return wrong low halves next to
    # the main stream's bf16 MFMA traffic (DESIGN.md rule 10; rtfs_net_amd/build.py builds those files without the SLP vectoriser since).
    # tools/repeat_small.py checks run-to-run bit identity over 150 forwards per mode; here the bound stays a tolerance.
    assert rel(outs[0], outs[3]) < (1e-7 if dtype == "f32" else 3e-5)
    # fp32: only the 1e-7-level regrouping of the statistics' partial sums; split-bf16: that perturbation re-rounds the hi / lo operand splits
    # downstream, i.e. the mode's own 2^-18 product error is re-drawn (tools/check_bf16_entries.py: 4.5e-6 per entry point)
    tol = 1e-6 if dtype == "f32" else 3e-5
    e31, e21 = rel(outs[3], outs[1]), rel(outs[2], outs[1])
    print(f"{dtype}: 8-wave vs two-workgroup form {e31:.2e}, 4-wave one-workgroup vs two-workgroup form {e21:.2e}")
    assert e31 < tol and e21 < tol


@pytest.mark.parametrize("B,T2", [(2, 125), (1, 300), (1, 625), (1, 1100)])
def test_attention_core_forward_and_adjoint_isolated(B, T2):
    """rtfs_attn_core_fwd (with the log-sum-exp output of the training step) + rtfs_attn_core_bwd against float64 autograd of
    softmax(Q K^T / 16) V (attention.py:171-173) on the GPU.  T2 = 625 (10 s): the adjoint walks keys / queries in two blocks of 512;
    T2 = 1100 (17.6 s): the forward is the key-blocked two-sweep kernel, whose final statistics supply the log-sum-exp."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(T2)
    Q = (torch.randn(B, 4, T2, 256, generator=g) * 0.5).cuda()
    K = (torch.randn(B, 4, T2, 256, generator=g) * 0.5).cuda()
    V = torch.randn(B, 4, T2, 1024, generator=g).cuda()
    dOh = torch.randn(B, 4, T2, 1024, generator=g).cuda()  # per head [T2][16 ch][64 f]
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (Q, K, V))
    ref = torch.softmax(q64 @ k64.transpose(-1, -2) / 16.0, -1) @ v64
    (ref * dOh.double()).sum().backward()
    # O layout of the kernels: [B][T2][64 ch = head * 16 + c][64 f]
    to_o = lambda t: t.view(B, 4, T2, 16, 64).permute(0, 2, 1, 3, 4).reshape(B, T2, 4096).contiguous()  # noqa: E731
    O = torch.empty(B * T2 * 4096, device="cuda")
    LSE = torch.empty(B * 4 * T2, device="cuda")
    lib.call("rtfs_attn_core_fwd", Q, K, V, O, LSE, B, T2)
    assert rel(O.view(B, T2, 4096), to_o(ref.detach().float())) < 2e-6
    lse_ref = torch.logsumexp(q64.detach() @ k64.detach().transpose(-1, -2) / 16.0, -1)
    assert float((LSE.view(B, 4, T2).double() - lse_ref).abs().max()) < 1e-4
    dO = to_o(dOh)
    dQ, dK, dV, D = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty(B * 4 * T2, device="cuda")
    lib.call("rtfs_attn_core_bwd", Q, K, V, O, dO, LSE, D, dQ, dK, dV, B, T2)
    assert rel(dQ, q64.grad) < 2e-5 and rel(dK, k64.grad) < 2e-5 and rel(dV, v64.grad) < 2e-5


@pytest.mark.parametrize("terms", [0, 1, 3, 6])
def test_attention_qkv_isolated_values_and_repeatability(terms):
    """rtfs_attn_qkv_fwd(_bf16) at the bench shape (32 x 125 tokens, two workgroups per CU) against a float64 restatement of the 12
    ConvActNorm modules (attention.py:30-66: 1x1 conv -> PReLU -> LN4D over (channel, F)), and bit-for-bit the same over 40 launches.
    Regression test for the packed-fp32 scatter: in the bf16 modes it sporadically wrote beta instead of the normalised value for 16
    consecutive float4 stores (zeros here, where beta = 0) - tools/qkv_det.py, attention.hip ln_apply."""
    from rtfs_net_amd import lib
    from rtfs_net_amd.models.hip_path import pack_bf16

    B, T2 = 32, 125
    g = torch.Generator(device="cuda").manual_seed(5)
    G = torch.randn(B * T2, 64, 64, device="cuda", generator=g)  # [token][f][channel]
    W = torch.randn(96, 64, device="cuda", generator=g) * 0.1
    bias = torch.randn(96, device="cuda", generator=g) * 0.1
    slope = torch.full((96,), 0.25, device="cuda")
    gq, bq = torch.rand(4, 256, device="cuda", generator=g) + 0.5, torch.zeros(4, 256, device="cuda")
    gv, bv = torch.rand(4, 1024, device="cuda", generator=g) + 0.5, torch.zeros(4, 1024, device="cuda")
    Wk = pack_bf16(W) if terms in (1, 3) else W
    name = "rtfs_attn_qkv_fwd" + ("_bf16" if terms else "")

    def launch():
        Q = torch.full((B, 4, T2, 256), float("nan"), device="cuda")
        K = torch.full_like(Q, float("nan"))
        V = torch.full((B, 4, T2, 1024), float("nan"), device="cuda")
        lib.call(name, G, Wk, bias, slope, gq, bq, gq, bq, gv, bv, Q, K, V, None, B, T2, *((terms,) if terms else ()))
        return Q, K, V

    first = launch()
    for _ in range(40):
        for a, b in zip(launch(), first):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # float64 restatement for the first 8 tokens of every utterance: y[f][n] = PReLU(G W^T + b); columns [0,16) Q, [16,32) K, [32,96) V
    tok = torch.arange(B * T2, device="cuda").view(B, T2)[:, :8].reshape(-1)
    y = G[tok].double() @ W.double().t() + bias.double()
    y = torch.where(y >= 0, y, y * slope.double())  # [tokens][f][96]

    def ln(cols, per_head, gamma):  # LN4D over (channel, f) per token and head; output [tokens][4][per_head * 64], index c * 64 + f
        z = y[:, :, cols].view(-1, 64, 4, per_head).permute(0, 2, 3, 1)  # [tokens][head][c][f]
        mu = z.mean((2, 3), keepdim=True)
        var = ((z - mu) ** 2).mean((2, 3), keepdim=True)
        return ((z - mu) / torch.sqrt(var + 1e-5)).reshape(-1, 4, per_head * 64) * gamma.double()

    tol = {0: 3e-6, 6: 3e-6, 3: 3e-5, 1: 2e-2}[terms]
    for out, cols, per_head, gamma in ((first[0], slice(0, 16), 4, gq), (first[1], slice(16, 32), 4, gq), (first[2], slice(32, 96), 16, gv)):
        got = out[:, :, :8].permute(0, 2, 1, 3).reshape(-1, 4, per_head * 64)
        assert not bool(torch.isnan(out).any())
        assert rel(got, ln(cols, per_head, gamma).float()) < tol


@pytest.mark.parametrize("terms,tol", [(0, 2e-5), (3, 5e-5), (1, 2e-2)])
def test_pixel_gemm_256_weight_stationary_form(terms, tol):
    """rtfs_bottleneck_fwd / rtfs_mask_fwd / rtfs_gemm_rows (256 -> 256; fp32, split-bf16 and bf16 MFMA) at a size that takes the weight-stationary kernel (ws256_kernel:
    >= 8192 32-pixel tiles), ragged last tile, against the oracle's own functions (conv_norm_act with pre gLN + ReLU, tdavnet.py:59,89;
    s3_mask, mask_generator.py:67-99) - and bit-identical to the LDS-staged kernel the small shapes take (same k order per accumulator)."""
    from oracle.avnet_ref import P, conv_norm_act, s3_mask
    from rtfs_net_amd import lib
    from rtfs_net_amd.models.hip_path import pack_bf16

    sfx, targs = ("_bf16", (terms,)) if terms else ("", ())

    g = torch.Generator().manual_seed(5)
    B, T = 33, 63  # 8127 pixels per utterance: 254 tiles, the last one with 31 rows
    TF = T * 129
    x = torch.randn(B, 256, T, 129, generator=g)
    emb = torch.randn(B, 256, T, 129, generator=g)
    W = torch.randn(256, 256, generator=g) * 0.06
    bias = torch.randn(256, generator=g) * 0.1
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    slope = torch.tensor([0.2])
    xcl, embcl = x.permute(0, 2, 3, 1).contiguous().cuda(), emb.permute(0, 2, 3, 1).contiguous().cuda()
    Wd, bd = (pack_bf16(W.cuda()) if terms else W.cuda()), bias.cuda()
    stats = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
    stats[:, 0], stats[:, 1] = xcl.double().sum((1, 2, 3)), (xcl.double() ** 2).sum((1, 2, 3))

    def nchw(y):
        return y.view(B, T, 129, 256).permute(0, 3, 1, 2).cpu()

    sub = slice(0, B, 8)  # the oracle on every 8th utterance (the kernel treats them alike; keeps the CPU side at seconds)
    # audio bottleneck
    a0 = torch.empty(B * TF * 256, device="cuda")
    lib.call("rtfs_bottleneck_fwd" + sfx, xcl, stats, gamma.cuda(), beta.cuda(), Wd, bd, a0, B, TF, *targs)
    sd = {"full_layer.0.norm.weight": gamma, "full_layer.0.norm.bias": beta, "full_layer.2.weight": W.view(256, 256, 1, 1), "full_layer.2.bias": bias}
    ref = conv_norm_act(x[sub], P(sd), is2d=True, pre_norm="gLN", pre_act="ReLU")
    assert rel(nchw(a0)[sub], ref) < tol
    # S3 mask (+ the post-ReLU mask output of the training step)
    for with_m in (False, True):
        masked = torch.full((B * TF * 256,), float("nan"), device="cuda")
        m = torch.full((B * TF * 256,), float("nan"), device="cuda") if with_m else None
        lib.call("rtfs_mask_fwd" + sfx, xcl, float(slope), Wd, bd, embcl, masked, m, B, TF, *targs)
        sdm = {"mask_generator.0.weight": slope, "mask_generator.1.full_layer.2.weight": W.view(256, 256, 1, 1), "mask_generator.1.full_layer.2.bias": bias}
        refm = s3_mask(x[sub], emb[sub], P(sdm), 1)[:, 0]
        assert not bool(torch.isnan(masked).any())
        assert rel(nchw(masked)[sub], refm) < tol
        if with_m:
            mref = F.relu(F.conv2d(F.prelu(x[sub], slope), W.view(256, 256, 1, 1), bias))
            assert rel(nchw(m)[sub], mref) < tol
    # plain rows GEMM (input-gradient GEMMs of the training step) and the small-shape kernel on a slice: identical bits
    y = torch.empty(B * TF * 256, device="cuda")
    lib.call("rtfs_gemm_rows" + sfx, xcl, Wd, None, y, B * TF, 256, 256, 0, *targs)
    assert rel(nchw(y)[sub], F.conv2d(x[sub], W.view(256, 256, 1, 1))) < tol
    n = 4096 + 17
    ys = torch.empty(n * 256, device="cuda")
    lib.call("rtfs_gemm_rows" + sfx, xcl.view(-1, 256)[:n].contiguous(), Wd, None, ys, n, 256, 256, 0, *targs)
    assert torch.equal(ys.view(torch.int32), y[: n * 256].view(torch.int32))


@pytest.mark.parametrize("terms,tol", [(0, 2e-6), (3, 2e-5), (1, 1e-2)])
@pytest.mark.parametrize("B,T2", [(17, 125), (9, 250), (3, 125)])
@pytest.mark.parametrize("dim", [4, 3])
def test_convt_entry_isolated(B, T2, dim, terms, tol):
    """rtfs_dp_convt_fwd in isolation (ConvTranspose1d(64 -> 64, k = 8) + bias + residual, in place; rnn_layers.py:129,153-156) against float64
    on the CPU: two sizes that take the weight-stationary kernel (>= 1536 64-row tiles: full / ragged time tiles, odd tile counts) and one
    that takes the LDS-staged kernel; fp32, split-bf16 and bf16 MFMA."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(7 * B + T2 + dim)
    S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
    L = npos - 7
    H3 = torch.randn(S, L, 64, generator=g)
    W = torch.randn(64, 512, generator=g) * 0.05  # [out channel][k' * 64 + in channel], k' = 7 - tap (the layout the host prepares)
    bias = torch.randn(64, generator=g) * 0.1
    G0 = torch.randn(B, T2, 64, 64, generator=g)
    G = G0.clone().cuda()
    if terms:
        from rtfs_net_amd.models.hip_path import pack_bf16

        lib.call("rtfs_dp_convt_fwd_bf16", H3.cuda(), pack_bf16(W.cuda()), bias.cuda(), G, B, T2, dim, terms)
    else:
        lib.call("rtfs_dp_convt_fwd", H3.cuda(), W.cuda(), bias.cuda(), G, B, T2, dim)
    hp = torch.zeros(S, npos + 14, 64, dtype=torch.float64)
    hp[:, 7:7 + L] = H3.double()
    win = torch.stack([hp[:, k:k + npos] for k in range(8)], dim=2).reshape(S, npos, 512)
    y = win @ W.double().t() + bias.double()
    y = y.view(B, T2, 64, 64) if dim == 4 else y.view(B, 64, T2, 64).permute(0, 2, 1, 3)
    assert rel(G.cpu(), y + G0.double()) < tol
    if terms == 0:
        # the two larger sizes take the fast-FIR kernel (pair rows per sequence 33 / 63 / 126: odd and even position counts, two- and three-sequence
        # tiles); form 1 = the direct 8-tap kernels.  Every output ROW on its own as well (a mis-addressed halo unit shows in single rows first)
        G1 = G0.clone().cuda()
        lib.call("rtfs_dp_convt_fwd_form", H3.cuda(), W.cuda(), bias.cuda(), G1, B, T2, dim, 1)
        assert rel(G1.cpu(), y + G0.double()) < tol and rel(G, G1) < 1e-6
        want = (y + G0.double()).reshape(-1, 64)
        rows = (G.double().cpu().reshape(-1, 64) - want).norm(dim=-1) / want.norm(dim=-1)
        assert float(rows.max()) < 5e-6, (float(rows.max()), int(rows.argmax()))
        with pytest.raises(RuntimeError):
            lib.call("rtfs_dp_convt_fwd_form", H3.cuda(), W.cuda(), bias.cuda(), G1, B, T2, dim, 2)
    # the out-of-place form (training step: the stage's input survives for the adjoint): the same bits as the in-place call, the input untouched
    Gin, Gout = G0.clone().cuda(), torch.full_like(G, float("nan"))
    if terms:
        lib.call("rtfs_dp_convt_fwd_to_bf16", H3.cuda(), pack_bf16(W.cuda()), bias.cuda(), Gin, Gout, B, T2, dim, terms)
    else:
        lib.call("rtfs_dp_convt_fwd_to", H3.cuda(), W.cuda(), bias.cuda(), Gin, Gout, B, T2, dim)
    assert torch.equal(Gout, G) and torch.equal(Gin.cpu(), G0)
    if not terms:  # Gin == Gout: the in-place call through the out-of-place entry point
        lib.call("rtfs_dp_convt_fwd_to", H3.cuda(), W.cuda(), bias.cuda(), Gin, Gin, B, T2, dim)
        assert torch.equal(Gin, G)


def test_attn_out_entry_in_place_and_out_of_place():
    """rtfs_attn_out_fwd (out-projection 64 -> 64 over the channels of every (token, frequency) + PReLU + LayerNorm over (64, F) + residual, attention.py:183-189)
    against float64, and its out-of-place form rtfs_attn_out_fwd_to (training step: the attention's input survives for the adjoint): the same bits, input untouched."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(5)
    B, T2 = 3, 21
    O = torch.randn(B * T2, 64, 64, generator=g)  # [token][c][f]
    W, bias, slope = torch.randn(64, 64, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1, 0.3
    gamma_fc, beta_fc = torch.rand(64, 64, generator=g) + 0.5, torch.randn(64, 64, generator=g) * 0.1  # [f][c]
    G0 = torch.randn(B * T2, 64, 64, generator=g)  # [token][f][c]
    y = torch.einsum("oc,tcf->tof", W.double(), O.double()) + bias.double()[None, :, None]
    y = torch.where(y >= 0, y, slope * y)
    mean, var = y.mean(dim=(1, 2), keepdim=True), y.var(dim=(1, 2), unbiased=False, keepdim=True)
    want = ((y - mean) / torch.sqrt(var + 1e-5)).permute(0, 2, 1) * gamma_fc.double() + beta_fc.double() + G0.double()
    G, ypre = G0.clone().cuda(), torch.empty(B * T2 * 4096, device="cuda")
    lib.call("rtfs_attn_out_fwd", O.cuda(), W.cuda(), bias.cuda(), slope, gamma_fc.cuda(), beta_fc.cuda(), G, ypre, B, T2)
    assert rel(G.cpu(), want) < 2e-6
    Gin, Gout, ypre2 = G0.clone().cuda(), torch.full_like(G, float("nan")), torch.empty_like(ypre)
    lib.call("rtfs_attn_out_fwd_to", O.cuda(), W.cuda(), bias.cuda(), slope, gamma_fc.cuda(), beta_fc.cuda(), Gin, Gout, ypre2, B, T2)
    assert torch.equal(Gout, G) and torch.equal(ypre2, ypre) and torch.equal(Gin.cpu(), G0)


@pytest.mark.parametrize("B,T2", [(19, 125), (10, 250), (3, 125), (23, 50)])
@pytest.mark.parametrize("dim", [4, 3])
def test_fold_gemm_bwd_entry_isolated(B, T2, dim):
    """rtfs_fold_gemm_bwd in isolation (input gradient of LN4D-output -> unfold -> layer-0 GEMM, autograd over rnn_layers.py:146-150:
    dxn[p][c] = sum_k sum_n dU0[p - k][n] W0[n][64 k + c]) against float64, every output row on its own as well (until round 5 this entry point was
    only covered end to end by the gradient fixtures)."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(11 * B + T2 + dim)
    S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
    L = npos - 7
    dU = torch.randn(S, L, 256, generator=g)
    W0 = torch.randn(256, 512, generator=g) * 0.05  # [n][64 k + c]
    Wf = W0.view(256, 8, 64).flip(1).permute(2, 1, 0).reshape(64, 2048).contiguous()  # [c][256 k' + n], k' = 7 - k (hip_train.TrainWeights)
    want = torch.zeros(S, npos, 64, dtype=torch.float64)
    for k in range(8):
        want[:, k:k + L] += dU.double() @ W0.double()[:, 64 * k:64 * k + 64]
    want = want.view(B, T2, 64, 64) if dim == 4 else want.view(B, 64, T2, 64).permute(0, 2, 1, 3)
    dxn = torch.full((B, T2, 64, 64), float("nan"), device="cuda")
    lib.call("rtfs_fold_gemm_bwd", dU.cuda(), Wf.cuda(), dxn, B, T2, dim)
    assert rel(dxn.cpu(), want) < 2e-6
    rows = (dxn.double().cpu() - want).reshape(-1, 64).norm(dim=-1) / want.reshape(-1, 64).norm(dim=-1)
    assert float(rows.max()) < 5e-6, (float(rows.max()), int(rows.argmax()))


@pytest.mark.parametrize("S,L", [(1, 1), (3, 7), (5, 8), (6, 9), (9, 33), (130, 57), (71, 118), (2100, 57), (4200, 20)])
def test_sru_layer_bwd_entry_matches_three_launches(S, L):
    """rtfs_sru_layer_bwd (recurrence adjoint + weight gradient + input gradient of an SRU layer 1-3 in one launch, dU held in LDS; autograd over
    sru.SRU's layer, rnn_layers.py:100-105) against the three launches it replaces - rtfs_sru_scan_bwd(km = 3), rtfs_wgrad, rtfs_gemm_rows - which the
    gradient fixtures pin to the float64 oracle: same dX (= dX0 + dX1, the two directions' parts), dW, dwc, dbias up to the order of the fp32 sums.
    Lengths on both sides of the 8-step chunk, odd and even sequence counts (a wave owns a pair), more pairs than one pass of the persistent grid;
    the incoming gradient in one part and split in two (dH + dH2, as the layer below a fused layer receives it)."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(97 * S + L)
    X = torch.randn(S, L, 64, generator=g).cuda()
    W = (torch.randn(192, 64, generator=g) * 0.15).cuda()
    wc = (torch.randn(128, generator=g) * 0.5).cuda()
    bias = (torch.randn(128, generator=g) * 0.5).cuda()
    dH = torch.randn(S, L, 64, generator=g).cuda()
    scale = 1.7
    H, C, U = torch.empty_like(X), torch.empty_like(X), torch.empty(S, L, 192, device="cuda")
    lib.call("rtfs_sru_layer_fwd", X, W, wc, bias, scale, H, C, U, S, L)
    # the three launches
    dU, dX0 = torch.empty_like(U), torch.empty_like(X)
    dwc0, db0, dW0 = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda"), torch.zeros(192 * 64, device="cuda")
    lib.call("rtfs_sru_scan_bwd", U, X, C, wc, bias, scale, dH, dU, dX0, dwc0, db0, S, L, 3)
    lib.call("rtfs_wgrad", dU, 192, X, 64, dW0, 64, None, S * L, 0, 0, 0, 1, 192, 64, 0, None, None, 0.0, None, 0)
    lib.call("rtfs_gemm_rows", dU, W.t().contiguous(), None, dX0, S * L, 192, 64, 1)
    part = torch.randn(S, L, 64, generator=g).cuda()
    for dHa, dHb in ((dH, None), (part, dH - part)):
        # one launch; the accumulated outputs start from a known non-zero value
        dXa, dXb = torch.full_like(X, float("nan")), torch.full_like(X, float("nan"))
        dwc1, db1, dW1 = torch.ones(128, device="cuda"), torch.ones(128, device="cuda"), torch.ones(192 * 64, device="cuda")
        work = torch.full((lib.load().rtfs_sru_layer_bwd_work_floats(S),), float("nan"), device="cuda")
        lib.call("rtfs_sru_layer_bwd", U, X, C, W, wc, bias, scale, dHa, dHb, dXa, dXb, work, dW1, dwc1, db1, S, L)
        torch.cuda.synchronize()
        assert torch.isfinite(dXa).all() and torch.isfinite(dXb).all()
        dX1 = dXa + dXb
        tol = 2e-6 if dHb is None else 2e-5  # (dH - part) + part is dH only to an ulp of the larger operand
        assert rel(dX1, dX0) < tol, rel(dX1, dX0)
        rows = (dX1.double() - dX0.double()).norm(dim=-1) / dX0.double().norm(dim=-1).clamp_min(1e-3)
        assert float(rows.max()) < 10 * tol, (float(rows.max()), int(rows.argmax()))
        assert rel(dW1 - 1, dW0) < 10 * tol and rel(dwc1 - 1, dwc0) < 10 * tol and rel(db1 - 1, db0) < 10 * tol, (
            rel(dW1 - 1, dW0), rel(dwc1 - 1, dwc0), rel(db1 - 1, db0))
    # rtfs_sru_scan_bwd2 (the layer below a fused layer: gradient in two parts) = rtfs_sru_scan_bwd on the sum
    dU2, dX2 = torch.empty_like(U), torch.empty_like(X)
    dwc2, db2 = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")
    lib.call("rtfs_sru_scan_bwd2", U, X, C, wc, bias, scale, part, dH - part, dU2, dX2, dwc2, db2, S, L, 3)
    assert rel(dU2, dU) < 2e-5 and rel(dwc2, dwc0) < 2e-4


@pytest.mark.parametrize("S,L", [(3, 7), (6, 9), (9, 33), (40, 57)])
def test_sru_layer_bwd_entry_against_float64_autograd(S, L):
    """rtfs_sru_layer_bwd on its own against float64 autograd over a plain restatement of the bidirectional SRU layer (sru.SRU with highway skip and
    weight_c, as oracle/sru_ref.py states it: U = X W^T; per direction f = sigmoid(u1 + b_f + w_f c), r = sigmoid(u2 + b_r + w_r c),
    c' = u0 + (c - u0) f, h = x' + (c' - x') r, x' = scale_x x; rnn_layers.py:100-105): dX (= dX0 + dX1), dW, dwc, dbias."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(53 * S + L)
    X = torch.randn(S, L, 64, generator=g)
    W = torch.randn(192, 64, generator=g) * 0.15
    wc, bias = torch.randn(128, generator=g) * 0.5, torch.randn(128, generator=g) * 0.5
    dH = torch.randn(S, L, 64, generator=g)
    scale = 1.7
    Xd, Wd, wcd, bd = (t.double().cuda().requires_grad_(True) for t in (X, W, wc, bias))
    U = (Xd @ Wd.t()).view(S, L, 3, 2, 32)  # column = gate * 64 + direction * 32 + unit
    xp = (Xd * scale).view(S, L, 2, 32)
    hs = []
    for d in (0, 1):
        c = torch.zeros(S, 32, dtype=torch.float64, device="cuda")
        out = [None] * L
        for t in (range(L) if d == 0 else range(L - 1, -1, -1)):
            u0, u1, u2 = U[:, t, 0, d], U[:, t, 1, d], U[:, t, 2, d]
            f = torch.sigmoid(u1 + bd[d * 32:d * 32 + 32] + wcd[d * 32:d * 32 + 32] * c)
            r = torch.sigmoid(u2 + bd[64 + d * 32:96 + d * 32] + wcd[64 + d * 32:96 + d * 32] * c)
            c = u0 + (c - u0) * f
            out[t] = xp[:, t, d] + (c - xp[:, t, d]) * r
        hs.append(torch.stack(out, dim=1))
    Hd = torch.cat(hs, dim=-1)  # [S][L][64], column = direction * 32 + unit
    (Hd * dH.double().cuda()).sum().backward()
    # the HIP path: forward (saves), then the one-launch adjoint
    Xc, Wc, wcc, bc, dHc = X.cuda(), W.cuda(), wc.cuda(), bias.cuda(), dH.cuda()
    H, C, Uc = torch.empty_like(Xc), torch.empty_like(Xc), torch.empty(S, L, 192, device="cuda")
    lib.call("rtfs_sru_layer_fwd", Xc, Wc, wcc, bc, scale, H, C, Uc, S, L)
    assert rel(H, Hd.detach()) < 2e-6
    dXa, dXb = torch.empty_like(Xc), torch.empty_like(Xc)
    dW, dwc, db = torch.zeros(192 * 64, device="cuda"), torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")
    work = torch.empty(lib.load().rtfs_sru_layer_bwd_work_floats(S), device="cuda")
    lib.call("rtfs_sru_layer_bwd", Uc, Xc, C, Wc, wcc, bc, scale, dHc, None, dXa, dXb, work, dW, dwc, db, S, L)
    assert rel(dXa + dXb, Xd.grad) < 5e-6, rel(dXa + dXb, Xd.grad)
    assert rel(dW.view(192, 64), Wd.grad) < 5e-6 and rel(dwc, wcd.grad) < 5e-6 and rel(db[:128], bd.grad) < 5e-6, (
        rel(dW.view(192, 64), Wd.grad), rel(dwc, wcd.grad), rel(db, bd.grad))


@pytest.mark.parametrize("B,T2", [(19, 125), (10, 250), (3, 125)])
@pytest.mark.parametrize("dim", [4, 3])
def test_convt_bwd_input_entry_isolated(B, T2, dim):
    """rtfs_convt_bwd_input in isolation (input gradient of ConvTranspose1d(64 -> 64, k = 8): dH3[l][j] = sum_{k, c} dG[l + k][c] Wt[j][64 k + c]; autograd of
    rnn_layers.py:153) against float64: two sizes that take the fast-FIR kernel (unfold_ffa_kernel<DIM, 2>, >= 1024 tiles of 63 pair rows) and one that takes the
    direct kernel; form 1 names the direct kernel at every size."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(11 * B + T2 + dim)
    S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
    L = npos - 7
    dG = torch.randn(B, T2, 64, 64, generator=g)
    W = torch.randn(64, 512, generator=g) * 0.05
    seqs = (dG if dim == 4 else dG.permute(0, 2, 1, 3)).reshape(S, npos, 64).double()
    win = torch.stack([seqs[:, k:k + L] for k in range(8)], dim=2).reshape(S, L, 512)
    want = win @ W.double().t()
    out = torch.full((S * L * 64,), float("nan"), device="cuda")
    lib.call("rtfs_convt_bwd_input", dG.cuda(), W.cuda(), out, B, T2, dim)
    out1 = torch.full_like(out, float("nan"))
    lib.call("rtfs_convt_bwd_input_form", dG.cuda(), W.cuda(), out1, B, T2, dim, 1)
    assert rel(out.view(want.shape), want) < 2e-6 and rel(out1.view(want.shape), want) < 2e-6 and rel(out, out1) < 1e-6
    rows = (out.view(want.shape).double().cpu() - want).norm(dim=-1) / want.norm(dim=-1)
    assert float(rows.max()) < 5e-6, (float(rows.max()), int(rows.argmax()))
    fast_fir = (S * ((L + 2) // 2) + 62) // 63 >= 1024
    assert fast_fir or torch.equal(out, out1)
    with pytest.raises(RuntimeError):
        lib.call("rtfs_convt_bwd_input_form", dG.cuda(), W.cuda(), out1, B, T2, dim, 2)


@pytest.mark.parametrize("kind,B,T2,dim", [("l0", 20, 125, 4), ("l0", 20, 125, 3), ("l0", 9, 250, 3), ("ct", 20, 125, 4), ("ct", 20, 125, 3), ("l0", 23, 50, 3), ("l0", 2, 125, 4)])
def test_toeplitz_wgrad_entry_isolated(kind, B, T2, dim):
    """rtfs_wgrad on the two Toeplitz maps of a DualPathRNN in isolation against float64 (round 5: until then these launches were only covered end to end):
    "l0" = the weight gradient of LN4D + unfold + SRU layer-0 GEMM (dW0[n][64 z + k] = sum_l dU0[l][n] xn[l + z][k], autograd of rnn_layers.py:146-150),
    "ct" = the ConvTranspose1d weight + bias gradient (x_off = -7, zero rows outside the sequence; rnn_layers.py:153).  Every (output channel, tap) row on its own."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(13 * B + T2 + dim + (kind == "ct"))
    S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
    L = npos - 7
    if kind == "l0":
        nout, seg, xseg, xoff = 256, L, npos, 0
    else:
        nout, seg, xseg, xoff = 64, npos, L, -7
    dY = torch.randn(S * seg, nout, generator=g)
    X = torch.randn(S * xseg, 64, generator=g)
    Xs = torch.zeros(S, xseg + 16, 64, dtype=torch.float64)  # padded by 8 on both sides: X[seq][p] at index p + 8
    Xs[:, 8:8 + xseg] = X.view(S, xseg, 64).double()
    dYs = dY.view(S, seg, nout).double()
    want = torch.stack([torch.einsum("sln,slk->nk", dYs, Xs[:, 8 + xoff + z:8 + xoff + z + seg]) for z in range(8)], 1).reshape(nout, 512)
    want_b = dYs.sum((0, 1))
    dW = torch.zeros(nout, 512, device="cuda")
    db = torch.zeros(nout, device="cuda") if kind == "ct" else None
    lib.call("rtfs_wgrad", dY.cuda(), nout, X.cuda(), 64, dW, 512, db, S * seg, seg, xseg, xoff, 8, nout, 64, 0, None, None, 0.0, None, 0)
    assert rel(dW, want) < 3e-6, rel(dW, want)
    rows = (dW.double().cpu() - want).view(nout, 8, 64).norm(dim=-1) / want.view(nout, 8, 64).norm(dim=-1)
    assert float(rows.max()) < 2e-5, float(rows.max())
    if db is not None:
        assert rel(db, want_b) < 3e-6, rel(db, want_b)


@pytest.mark.parametrize("M", [228000, 241664, 65536 + 13, 40000])
def test_wgrad_sru_layer_map_isolated(M):
    """rtfs_wgrad on the SRU layer 1-3 map (dW[192][64] += dU^T . h: 36 launches per training step) in isolation against float64 (ragged row counts; dW is
    ACCUMULATED into).  (Round 5 built a one-workgroup-per-output kernel for this map and measured it no faster than the generic one - 81 against 77 us: with the
    row index as contraction both sides are read 4 bytes at a time from LDS and the MFMA and memory phases do not overlap - so it is not in the library.)"""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(M)
    dU = torch.randn(M, 192, generator=g)
    h = torch.randn(M, 64, generator=g)
    dW0 = torch.randn(192, 64, generator=g)
    dW = dW0.clone().cuda()
    lib.call("rtfs_wgrad", dU.cuda(), 192, h.cuda(), 64, dW, 64, None, M, 0, 0, 0, 1, 192, 64, 0, None, None, 0.0, None, 0)
    want = dU.double().t() @ h.double()
    assert rel(dW.double().cpu() - dW0.double(), want) < 3e-6
    rows = (dW.double().cpu() - dW0.double() - want).norm(dim=-1) / want.norm(dim=-1)
    assert float(rows.max()) < 2e-5


@pytest.mark.parametrize("M,K,acc", [(228000, 192, 1), (65537, 192, 1), (70001, 192, 0), (200000, 256, 0), (65536 + 31, 256, 0), (40000, 192, 1), (100003, 96, 1), (100003, 64, 0)])
def test_gemm_rows_narrow_maps_isolated(M, K, acc):
    """rtfs_gemm_rows onto 64 columns at the training step's sizes (dx += dU . W of the SRU layers: K = 192, accumulating; the residual conv's input gradient:
    K = 256) against float64: from 65536 rows up the weight-stationary 64-column kernel (rows_ws64_kernel, round 5; ragged last 32-row tile), below it the generic one."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(M + K + acc)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(64, K, generator=g) * 0.1
    Y0 = torch.randn(M, 64, generator=g)
    Y = torch.cat([Y0, torch.full((64, 64), 7.0)], 0).cuda()  # 64 guard rows behind the matrix: a store past row M would show
    lib.call("rtfs_gemm_rows", X.cuda(), W.cuda(), None, Y, M, K, 64, acc)
    want = X.double() @ W.double().t() + (Y0.double() if acc else 0)
    assert rel(Y[:M], want) < 1e-6
    rows = (Y[:M].double().cpu() - want).norm(dim=-1) / want.norm(dim=-1)
    assert float(rows.max()) < 1e-5 and bool((Y[M:] == 7.0).all())


def test_weight_stationary_kernels_in_the_model():
    """RTFS-Net-2 at the bench shape (batch 32, 2 s): the forward takes the weight-stationary kernels (256 -> 256 pixel GEMMs, layer-0 GEMM,
    ConvTranspose GEMM) and the one-workgroup residual kernels.  (1) with the layer-0 GEMM forced to the LDS-staged kernel (variant 2) the
    waveforms agree to 1e-6 (v_rsq in the weight-stationary kernel's LayerNorm); (2) every 8th utterance separated ALONE - a batch of one
    takes the small-batch kernel of every stage - agrees with its row of the batch to 1e-5."""
    model, sd, cfg = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(32, 32000, 50)
    mix, emb = mix.cuda(), emb.cuda()
    with torch.no_grad():
        out = model(mix, emb)
        model._hip.variants["unfold"] = 2
        try:
            staged = model(mix, emb)
        finally:
            model._hip.variants["unfold"] = 0
        assert torch.isfinite(out).all()
        assert rel(out, staged) < 1e-6
        for j in range(0, 32, 8):
            solo = model(mix[j:j + 1], emb[j:j + 1])
            assert rel(solo[0], out[j]) < 1e-5, j


@pytest.mark.parametrize("terms,tol", [(3, 3e-5), (1, 2e-2)])
@pytest.mark.parametrize("dim", [4, 3])
def test_unfold_gemm_bf16_weight_stationary_form(dim, terms, tol):
    """rtfs_dp_unfold_gemm_fwd_bf16 at a size that takes the weight-stationary kernel (operand tuples read whole from a re-ordered slab row, weight
    tuples regrouped once): against float64 within the mode's own error, and against the LDS-staged kernel of the same mode (variant 2) to the
    level at which the two LayerNorm forms (v_rsq / IEEE) re-round the bf16 splits."""
    from rtfs_net_amd import lib
    from rtfs_net_amd.models.hip_path import pack_bf16

    B, T2 = 10, 125
    g = torch.Generator().manual_seed(40 + dim + terms)
    G = torch.randn(B, T2, 64, 64, generator=g)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    Wt = torch.randn(256, 512, generator=g) * 0.05
    x = G.double()
    xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    seqs = xn.reshape(B * T2, 64, 64) if dim == 4 else xn.permute(0, 2, 1, 3).reshape(B * 64, T2, 64)
    L = seqs.shape[1] - 7
    win = torch.stack([seqs[:, k:k + L] for k in range(8)], dim=2).reshape(seqs.shape[0], L, 512)
    want = win @ Wt.double().t()
    Wk = pack_bf16(Wt.cuda())
    outs = []
    for variant in (0, 2):
        U = torch.full((seqs.shape[0] * L * 256,), float("nan"), device="cuda")
        lib.call("rtfs_dp_unfold_gemm_fwd_bf16", G.cuda(), gamma.cuda(), beta.cuda(), Wk, U, B, T2, dim, variant, terms)
        assert rel(U.view(want.shape), want) < tol
        outs.append(U)
    assert rel(outs[0], outs[1]) < tol


def test_deferred_finish_of_the_gradient_reducers():
    """rtfs_spread_defer (csrc/spread.hip): inside a deferred section the reducers' finish launches are recorded and applied at the flush - with
    atomic adds, because two producers may name the SAME destination (the RTFS blocks share their weights).  Thirty producers (more than one flush
    batch of 24), two destinations used 15 times each, against the immediate mode; rtfs_colsum_add's wide regions exercise the region cursor."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(3)
    n = 1 << 16
    dy, x = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    X = torch.randn(4096, 256, generator=g).cuda()
    dx = torch.empty(n, device="cuda")

    def run(deferred):
        ds = [torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")]
        cs = torch.zeros(256, device="cuda")
        if deferred:
            lib.spread_defer(True, "cuda:0")
        for k in range(30):
            lib.call("rtfs_prelu_bwd", dy, x, 0.25 + 0.01 * k, dx, 0, ds[k & 1], n)
            if k % 5 == 0:
                lib.call("rtfs_colsum_add", X, cs, 4096, 256)
        if deferred:
            assert float(ds[0].abs() + ds[1].abs()) >= 0  # (no flush forced here: the values are only complete after the section)
            lib.spread_defer(False, "cuda:0")
        torch.cuda.synchronize()
        return ds[0].clone(), ds[1].clone(), cs.clone()

    a0, a1, ac = run(False)
    b0, b1, bc = run(True)
    assert float(a0.abs()) > 0 and float(a1.abs()) > 0
    assert rel(b0, a0) < 1e-5 and rel(b1, a1) < 1e-5 and rel(bc, ac) < 1e-5
    ref = 6 * X.double().sum(0)
    assert rel(bc, ref) < 1e-5
    c0, c1, cc = run(False)  # immediate mode again after a deferred section: the scratch came back zeroed
    assert rel(c0, a0) < 1e-5 and rel(cc, ac) < 1e-5


@pytest.mark.parametrize("B,L", [(2, 16000 + 77), (1, 4096)])
def test_gadd_fusion_matches_separate_calls(B, L):
    """rtfs_dwconv_gadd_fwd (G = pooled + gLN(D1) formed inside the pass that computes fusion_layers[1]'s local embedding of gLN(D1)) against
    rtfs_pool_add_fwd + rtfs_dwconv_fwd, end to end (even / odd frame counts, ragged tiles)"""
    model, _, _ = make_model(2, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, max(8, L // 640))
    with torch.no_grad():
        fused = model(mix.cuda(), emb.cuda())
        model._hip.fuse["gadd"] = False
        plain = model(mix.cuda(), emb.cuda())
    assert rel(fused, plain) < 2e-6


@pytest.mark.parametrize("B,T,F,Tg,Fg", [(2, 51, 129, 13, 33), (3, 18, 129, 5, 33), (2, 13, 33, 13, 33), (1, 201, 129, 51, 33)])
def test_mix_gln_bwd_fusion_matches_separate_calls(B, T, F, Tg, Fg):
    """rtfs_mix_gln_bwd (adjoint of the InjectionMultiSum mix, fusion.py:59-67, with the gLN adjoint of its local branch folded in - no dNloc
    tensor - and the reduce passes of the gate / global branches' gLN adjoints riding along) against rtfs_mix_bwd + three rtfs_gln_bwd_reduce /
    _apply pairs, and both against float64 autograd of the formula, at an up-sampling footprint (ragged 129 -> 33, 51 -> 13) and at equal
    resolutions (fusion_layers[1])."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(11)
    H = 64
    loc, dOut = torch.randn(B, T, F, H, generator=g) * 1.5 + 0.3, torch.randn(B, T, F, H, generator=g)
    gate, glob = torch.randn(B, Tg, Fg, H, generator=g) * 0.7 - 0.2, torch.randn(B, Tg, Fg, H, generator=g) * 2.0 + 0.5
    gam = [torch.rand(H, generator=g) + 0.5 for _ in range(3)]
    bet = [torch.randn(H, generator=g) * 0.2 for _ in range(3)]

    def stats(x):
        s = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64)
        s[:, 0], s[:, 1] = x.double().flatten(1).sum(1), x.double().pow(2).flatten(1).sum(1)
        return s.cuda()

    # float64 autograd of the formula: leaves are the three PRE-NORM tensors and the three (gamma, beta) pairs
    x64 = [t.double().requires_grad_(True) for t in (loc, gate, glob)]
    ga64, be64 = [t.double().requires_grad_(True) for t in gam], [t.double().requires_grad_(True) for t in bet]

    def gln(x, ga, be):
        m, v = x.flatten(1).mean(1).view(-1, 1, 1, 1), x.flatten(1).var(1, unbiased=False).view(-1, 1, 1, 1)
        return (x - m) / torch.sqrt(v + 1e-8) * ga + be

    up = lambda x: torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(T, F), mode="nearest").permute(0, 2, 3, 1)  # noqa: E731
    out = gln(x64[0], ga64[0], be64[0]) * torch.sigmoid(up(gln(x64[1], ga64[1], be64[1]))) + up(gln(x64[2], ga64[2], be64[2]))
    (out * dOut.double()).sum().backward()

    d = lambda t: t.cuda().contiguous()  # noqa: E731
    xd = [d(loc), d(gate), d(glob)]
    sts = [stats(loc), stats(gate), stats(glob)]
    gd, bd = [d(t) for t in gam], [d(t) for t in bet]
    rows = [T * F, Tg * Fg, Tg * Fg]
    res = {}
    for fused in (True, False):
        dX = [torch.empty_like(t) for t in xd]
        dN = [torch.empty_like(t) for t in xd]
        dgb = [torch.zeros(H, device="cuda") for _ in range(6)]
        red = torch.zeros(3, B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
        if fused:
            lib.call("rtfs_mix_gln_bwd", d(dOut), xd[0], sts[0], gd[0], bd[0], xd[1], sts[1], gd[1], bd[1], xd[2], sts[2], gd[2], bd[2], dX[0], dN[1], dN[2],
                     red, dgb, B, T, F, Tg, Fg)
        else:
            lib.call("rtfs_mix_bwd", d(dOut), xd[0], sts[0], gd[0], bd[0], xd[1], sts[1], gd[1], bd[1], dN[0], dN[1], dN[2], B, T, F, Tg, Fg)
            for i in range(3):
                lib.call("rtfs_gln_bwd_reduce", dN[i], xd[i], sts[i], gd[i], bd[i], 0, 0.0, red[i], dgb[2 * i], dgb[2 * i + 1], None, B, rows[i], H)
            lib.call("rtfs_gln_bwd_apply", dN[0], xd[0], sts[0], gd[0], bd[0], 0, 0.0, red[0], dX[0], 0, B, rows[0], H)
        for i in (1, 2):
            lib.call("rtfs_gln_bwd_apply", dN[i], xd[i], sts[i], gd[i], bd[i], 0, 0.0, red[i], dX[i], 0, B, rows[i], H)
        torch.cuda.synchronize()
        res[fused] = dX + dgb
        for i, name in enumerate(("loc", "gate", "glob")):
            for got, want, what in ((dX[i], x64[i].grad, "dX"), (dgb[2 * i], ga64[i].grad, "dgamma"), (dgb[2 * i + 1], be64[i].grad, "dbeta")):
                assert rel(got.cpu().double(), want) < 3e-5, (fused, name, what, rel(got.cpu().double(), want))
    for a, b in zip(res[True], res[False]):
        assert rel(a, b) < 1e-5


@pytest.mark.parametrize("B,T,F,stride,mode", [(2, 51, 129, 1, 0), (2, 37, 129, 1, 1), (1, 70, 129, 1, 2), (2, 13, 33, 1, 0), (3, 18, 129, 2, 1),
                                               (2, 51, 129, 2, 2), (1, 33, 64, 1, 1)])
def test_dwconv_adjoints_match_float64_autograd(B, T, F, stride, mode):
    """rtfs_dwconv_bwd_weight (two output rows per thread at stride 1) and rtfs_dwconv_bwd_input against float64 autograd of the depth-wise 4x4
    convolution as ConvNormAct builds it (conv_layers.py:104-113: 'same' padding 1 / 2 at stride 1, padding 1 at stride 2, applied to the
    TRANSFORMED input: raw / gLN / PReLU(gLN)), ragged row counts (not multiples of the 32-row workgroup) and frequency segments."""
    import torch.nn.functional as Fn

    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(5)
    H = 64
    x = torch.randn(B, T, F, H, generator=g) * 1.3 + 0.2
    w = torch.randn(16, H, generator=g) * 0.2
    gam, bet = torch.rand(H, generator=g) + 0.5, torch.randn(H, generator=g) * 0.2
    slope = 0.25
    To, Fo = (T, F) if stride == 1 else ((T - 2) // 2 + 1, (F - 2) // 2 + 1)
    dOut = torch.randn(B, To, Fo, H, generator=g)

    x64 = x.double()
    if mode >= 1:
        m, v = x64.flatten(1).mean(1).view(-1, 1, 1, 1), x64.flatten(1).var(1, unbiased=False).view(-1, 1, 1, 1)
        xin = (x64 - m) / torch.sqrt(v + 1e-8) * gam.double() + bet.double()
        if mode == 2:
            xin = torch.where(xin >= 0, xin, slope * xin)
    else:
        xin = x64
    xin = xin.detach().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    bias64 = torch.zeros(H, dtype=torch.float64, requires_grad=True)
    xc = xin.permute(0, 3, 1, 2)
    xc = Fn.pad(xc, (1, 2, 1, 2)) if stride == 1 else Fn.pad(xc, (1, 1, 1, 1))
    out = Fn.conv2d(xc, w64.t().reshape(H, 1, 4, 4), bias64, stride=stride, groups=H).permute(0, 2, 3, 1)
    assert out.shape == dOut.shape
    (out * dOut.double()).sum().backward()

    st = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64)
    st[:, 0], st[:, 1] = x64.flatten(1).sum(1), x64.pow(2).flatten(1).sum(1)
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    dW, dbias = torch.zeros(16 * H, device="cuda"), torch.zeros(H, device="cuda")
    if mode == 0:
        lib.call("rtfs_dwconv_bwd_weight", d(dOut), d(x), None, None, None, 0.0, 0, stride, dW, dbias, B, T, F)
    else:
        lib.call("rtfs_dwconv_bwd_weight", d(dOut), d(x), d(st), d(gam), d(bet), slope, mode, stride, dW, dbias, B, T, F)
    dIn = torch.empty(B, T, F, H, device="cuda")
    lib.call("rtfs_dwconv_bwd_input", d(dOut), d(w), dIn, 0, stride, B, T, F)
    dIn2 = torch.ones(B, T, F, H, device="cuda")
    lib.call("rtfs_dwconv_bwd_input", d(dOut), d(w), dIn2, 1, stride, B, T, F)
    torch.cuda.synchronize()
    assert rel(dW.view(16, H).cpu().double(), w64.grad) < 1e-5
    assert rel(dbias.cpu().double(), bias64.grad) < 1e-5
    assert rel(dIn.cpu().double(), xin.grad) < 1e-5
    assert rel((dIn2 - 1).cpu().double(), xin.grad) < 1e-5


@pytest.mark.parametrize("B,T", [(2, 51), (3, 18), (1, 126), (2, 33)])
def test_d0_tail_bwd_matches_separate_calls(B, T):
    """rtfs_d0_tail_bwd (stride-2 conv input gradient + pooling adjoint + the reduce pass of D0's gLN adjoint in one pass over d(gLN(D0)):
    adjoint of tdanet.py:112-118) against rtfs_dwconv_bwd_input + rtfs_pool_bwd + rtfs_gln_bwd_reduce, odd / even frame counts."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(21)
    H, F, F2 = 64, 129, 64
    T2 = (T - 2) // 2 + 1
    D0 = (torch.randn(B, T, F, H, generator=g) * 1.4 + 0.3).cuda()
    dN0 = torch.randn(B, T, F, H, generator=g).cuda()
    dD1, dG = torch.randn(B, T2, F2, H, generator=g).cuda(), torch.randn(B, T2, F2, H, generator=g).cuda()
    w = (torch.randn(16, H, generator=g) * 0.2).cuda()
    gam, bet = (torch.rand(H, generator=g) + 0.5).cuda(), (torch.randn(H, generator=g) * 0.2).cuda()
    st = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
    st[:, 0], st[:, 1] = D0.double().flatten(1).sum(1), D0.double().pow(2).flatten(1).sum(1)
    res = {}
    for fused in (True, False):
        acc = dN0.clone()
        red = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
        dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
        if fused:
            lib.call("rtfs_d0_tail_bwd", dD1, w, dG, acc, D0, st, gam, bet, red, dg, db, B, T, T2)
        else:
            lib.call("rtfs_pool_bwd", dG, acc, B, T, T2)
            lib.call("rtfs_dwconv_bwd_input", dD1, w, acc, 1, 2, B, T, F)
            lib.call("rtfs_gln_bwd_reduce", acc, D0, st, gam, bet, 0, 0.0, red, dg, db, None, B, T * F, H)
        torch.cuda.synchronize()
        res[fused] = (acc, red[:, :2].clone(), dg, db)
    for a, b, name in zip(res[True], res[False], ("dN0", "S1/S2", "dgamma", "dbeta")):
        assert rel(a, b) < 1e-5, (name, rel(a, b))
    assert float((res[True][0] - dN0).abs().max()) > 0.1  # (something was added)


@pytest.mark.parametrize("switch", ["mixgln", "d0tail", "wgside", "decmask", "srubwd", "dwadj", "da0sum", "actepi", "enctail", "wgdefer", "nextde"])
def test_training_step_fusion_switches_leave_the_gradients_alone(switch):
    """The backward chain's host-side choices (models/hip_train.py: rtfs_mix_gln_bwd, rtfs_d0_tail_bwd, the weight-gradient launches on a side
    stream with their own scratch lane, the SRU layers' adjoint in one launch - rtfs_sru_layer_bwd - against rtfs_sru_scan_bwd2 + rtfs_wgrad + rtfs_gemm_rows, rtfs_dw_adjoint / rtfs_dw_adjoint_mix against the per-convolution launches, d(a0) summed once by rtfs_sum_n against the per-block read-modify-write) against the separate / in-line launches they replace, on a whole training step of RTFS-Net-3 (eval mode under autograd = the training-step path without dropout masks)."""
    model, _, _ = make_model(3, "cuda")
    model.eval()
    mix, _, emb = synth.synth_inputs(2, 6000, 12)
    wgt = torch.randn(2, 1, 6000, generator=torch.Generator().manual_seed(1)).cuda()
    grads = {}
    for on in (True, False):
        model._hip.fuse[switch] = on
        model.zero_grad(set_to_none=True)
        (model(mix.cuda(), emb.cuda()) * wgt).sum().backward()
        grads[on] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model._hip.fuse[switch] = True
    scale = max(float(g.norm()) for g in grads[True].values())
    for n, g in grads[True].items():
        assert float((g - grads[False][n]).norm()) <= 2e-5 * float(g.norm()) + 1e-6 * scale, n


def test_spread_scratch_first_use_on_a_side_stream_is_ordered_after_its_zero_fill():
    """A lane's scratch is allocated and zero-filled on its first use.  The fill runs on the NULL stream; a torch side stream is non-blocking, so
    nothing orders it against the lane's first producer unless the library waits for the fill.  In a fresh process: the null stream is kept busy, then
    a deferred section on lane 1 runs one reducer (rtfs_ln4d_c_bwd: dgamma, dbeta through the scratch) on a side stream and is flushed only after the
    null stream has drained - a fill that was still queued behind the busy null stream would wipe the pending partial sums (round 5: one parameter
    gradient of a process's first training step missing, once in ~25 runs of this suite)."""
    import subprocess
    import sys

    code = r"""
import sys
import torch
sys.path.insert(0, ".")
from rtfs_net_amd import lib
g = torch.Generator().manual_seed(3)
rows = 4096
dxn, G = torch.randn(rows, 64, generator=g).cuda(), torch.randn(rows, 64, generator=g).cuda()
gamma = (torch.rand(64, generator=g) + 0.5).cuda()
dG, dgam, dbet = torch.zeros(rows, 64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
x = torch.randn(8192, 8192, device="cuda")
side = torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(40):
    y = x @ x  # ~100 ms of work on the null stream
lib.spread_lane(1)
with torch.cuda.stream(side):
    lib.spread_defer(True, "cuda:0")
    lib.call("rtfs_ln4d_c_bwd", dxn, G, gamma, dG, dgam, dbet, rows)
    side.synchronize()
    torch.cuda.default_stream().synchronize()  # (a fill still queued on the null stream has run by now)
    lib.spread_defer(False, "cuda:0")
lib.spread_lane(0)
torch.cuda.synchronize()
Gd = G.double()
xh = (Gd - Gd.mean(-1, keepdim=True)) / torch.sqrt(Gd.var(-1, unbiased=False, keepdim=True) + 1e-5)
want_b, want_g = dxn.double().sum(0), (dxn.double() * xh).sum(0)
eb = float((dbet.double() - want_b).norm() / want_b.norm())
eg = float((dgam.double() - want_g).norm() / want_g.norm())
print("ERR", eb, eg)
assert eb < 1e-5 and eg < 1e-4, (eb, eg)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_spread_lanes_keep_concurrent_reducers_apart():
    """rtfs_spread_lane (csrc/spread.hip): reducers issued on a second stream with lane 1 run concurrently with lane 0's on the main stream - each lane has
    its own scratch, cursor and deferred section.  Forty producers per stream, interleaved, immediate and deferred mode, against a serial run; without the
    lanes the two streams' partial sums meet in one scratch region (the training step's weight-gradient side stream relies on this)."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(9)
    n = 1 << 20
    dy = [torch.randn(n, generator=g).cuda() for _ in range(2)]
    x = [torch.randn(n, generator=g).cuda() for _ in range(2)]
    side = torch.cuda.Stream()

    def run(two_streams, deferred):
        ds = [torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")]
        dx = [torch.empty(n, device="cuda"), torch.empty(n, device="cuda")]
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        if deferred:
            lib.spread_defer(True, "cuda:0")
            if two_streams:
                lib.spread_lane(1)
                with torch.cuda.stream(side):
                    lib.spread_defer(True, "cuda:0")
                lib.spread_lane(0)
        for k in range(40):
            lib.call("rtfs_prelu_bwd", dy[0], x[0], 0.25, dx[0], 0, ds[0], n)
            if two_streams:
                lib.spread_lane(1)
                with torch.cuda.stream(side):
                    lib.call("rtfs_prelu_bwd", dy[1], x[1], 0.5, dx[1], 0, ds[1], n)
                lib.spread_lane(0)
            else:
                lib.call("rtfs_prelu_bwd", dy[1], x[1], 0.5, dx[1], 0, ds[1], n)
        if deferred:
            if two_streams:
                lib.spread_lane(1)
                with torch.cuda.stream(side):
                    lib.spread_defer(False, "cuda:0")
                lib.spread_lane(0)
            lib.spread_defer(False, "cuda:0")
        main.wait_stream(side)
        torch.cuda.synchronize()
        return float(ds[0]), float(ds[1])

    want = run(False, False)
    assert abs(want[0]) > 1 and abs(want[1]) > 1
    for deferred in (False, True):
        got = run(True, deferred)
        assert abs(got[0] - want[0]) <= 1e-4 * abs(want[0]) and abs(got[1] - want[1]) <= 1e-4 * abs(want[1]), (deferred, got, want)
    again = run(False, True)  # both lanes came back clean
    assert abs(again[0] - want[0]) <= 1e-4 * abs(want[0]) and abs(again[1] - want[1]) <= 1e-4 * abs(want[1])


@pytest.mark.parametrize("rows", [3 * 18 * 129, 64 * 129 + 7])
def test_decoder_mask_bwd_matches_separate_calls(rows):
    """rtfs_decoder_mask_bwd (the S3 mask's element-wise adjoint in the paired epilogue of the decoder input-gradient GEMM; mask_generator.py:70-82
    behind decoder.py's ConvTranspose2d) against rtfs_gemm_rows + rtfs_mask_bwd_elem and against float64, ragged last tile."""
    from rtfs_net_amd import lib

    g = torch.Generator(device="cuda").manual_seed(12)
    dtaps = torch.randn(rows, 32, device="cuda", generator=g)
    wT = torch.randn(256, 32, device="cuda", generator=g) / 6
    a_emb = torch.randn(rows, 256, device="cuda", generator=g)
    m = torch.relu(torch.randn(rows, 256, device="cuda", generator=g))
    dz, de = torch.empty(rows, 256, device="cuda"), torch.empty(rows, 256, device="cuda")
    lib.call("rtfs_decoder_mask_bwd", dtaps, wT, a_emb, m, dz, de, rows)
    dm, dz2, de2 = torch.empty(rows, 256, device="cuda"), torch.empty(rows, 256, device="cuda"), torch.empty(rows, 256, device="cuda")
    lib.call("rtfs_gemm_rows", dtaps, wT, None, dm, rows, 32, 256, 0)
    lib.call("rtfs_mask_bwd_elem", dm, a_emb, m, dz2, de2, rows)
    torch.cuda.synchronize()
    assert torch.equal(dz, dz2) and torch.equal(de, de2)  # same products, same order
    d64 = dtaps.double() @ wT.double().t()
    dor, doi, er, ei, mr, mi = d64[:, :128], d64[:, 128:], a_emb.double()[:, :128], a_emb.double()[:, 128:], m.double()[:, :128], m.double()[:, 128:]
    want_dz = torch.cat([(dor * er + doi * ei) * (mr > 0), (doi * er - dor * ei) * (mi > 0)], 1)
    want_de = torch.cat([dor * mr + doi * mi, doi * mr - dor * mi], 1)
    assert rel(dz.double(), want_dz) < 1e-5 and rel(de.double(), want_de) < 1e-5
