"""GPU: rtfs_dw_adjoint (csrc/bwd_dw.hip) - the whole adjoint of 1 / 2 / 4 stride-1 depth-wise 4x4 convolutions that share an input, one launch - against float64
autograd of the forward it is the adjoint of (conv_layers.py:65-129 with groups = channels, 'same' padding of the even kernel; GroupNorm(1, C) behind each
convolution when the gLN adjoint rides on the load), through the C-ABI, isolated from the rest of the step.  Every output: gradient w.r.t. the transformed
input (plain and accumulated), tap gradients, bias gradients.  Ragged shapes: rows / columns that are no multiple of the 8 x 8 tile, several f segments."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
H = 64


def _cl(x):  # [B, 64, T, F] -> channels-last flat
    return x.permute(0, 2, 3, 1).contiguous().float()


def _case(nconv, gln, mode, accumulate, bias, B, T, Fq, seed):
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    x_in = rnd(B, H, T, Fq)
    in_g, in_b, slope = 1 + 0.3 * rnd(H), 0.2 * rnd(H), 0.25
    taps = [0.3 * rnd(16, H) for _ in range(nconv)]
    biases = [0.1 * rnd(H) for _ in range(nconv)] if bias else [None] * nconv
    gam, bet = [1 + 0.3 * rnd(H) for _ in range(nconv)], [0.1 * rnd(H) for _ in range(nconv)]
    dys = [rnd(B, H, T, Fq) for _ in range(nconv)]
    # float64 reference
    xin = x_in
    if mode >= 1:
        xin = F.group_norm(x_in, 1, in_g, in_b, 1e-5)
    if mode == 2:
        xin = F.prelu(xin, torch.tensor([slope], dtype=torch.float64))
    xin = xin.detach().requires_grad_(True)
    ws = [t.t().reshape(H, 1, 4, 4).clone().requires_grad_(True) for t in taps]
    bs = [b.clone().requires_grad_(True) if b is not None else None for b in biases]
    ys, loss = [], 0
    for k in range(nconv):
        y = F.conv2d(F.pad(xin, (1, 2, 1, 2)), ws[k], bs[k], groups=H)
        ys.append(y.detach())
        n = F.group_norm(y, 1, gam[k], bet[k], 1e-5) if gln else y
        loss = loss + (n * dys[k]).sum()
    loss.backward()
    # HIP
    dev = "cuda"
    N = T * Fq * H
    slot = lambda t: torch.stack([t.reshape(B, -1).sum(1), (t.reshape(B, -1) ** 2).sum(1)] + [torch.zeros(B, dtype=torch.float64)] * (lib.STAT_STRIDE - 2), 1).contiguous().to(dev)  # noqa: E731
    x_cl = [_cl(y).to(dev) for y in ys] if gln else None
    x_st = [slot(y.float().double()) for y in ys] if gln else None
    red = None
    if gln:
        red = []
        for k in range(nconv):
            y32 = ys[k].float().double()
            mean = y32.reshape(B, -1).mean(1).view(B, 1, 1, 1)
            var = (y32.reshape(B, -1) ** 2).mean(1).view(B, 1, 1, 1) - mean ** 2
            xh = (y32 - mean) / torch.sqrt(var + 1e-5)
            a = dys[k] * gam[k].view(1, H, 1, 1)
            r = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64)
            r[:, 0], r[:, 1] = a.reshape(B, -1).sum(1), (a * xh).reshape(B, -1).sum(1)
            red.append(r.to(dev))
    dIn0 = torch.randn(B * N, generator=torch.Generator().manual_seed(seed + 1)).to(dev) if accumulate else torch.full((B * N,), float("nan"), device=dev)
    dIn = dIn0.clone()
    dW = [torch.zeros(16 * H, device=dev) for _ in range(nconv)]
    db = [torch.zeros(H, device=dev) for _ in range(nconv)] if bias else None
    f32 = lambda t: t.float().contiguous().to(dev)  # noqa: E731
    lib.call("rtfs_dw_adjoint", nconv, [_cl(d).to(dev) for d in dys], x_cl, x_st, red, [f32(t) for t in gam] if gln else None, [f32(t) for t in taps],
             _cl(x_in).to(dev), slot(x_in.float().double()) if mode >= 1 else None, f32(in_g) if mode >= 1 else None, f32(in_b) if mode >= 1 else None,
             slope, mode, None, 0, 0, dIn, 1 if accumulate else 0, dW, db, B, T, Fq)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())  # noqa: E731
    ref_dIn = _cl(xin.grad).double().reshape(-1) + (dIn0.double().cpu() if accumulate else 0)
    errs = {"dIn": rel(dIn, ref_dIn)}
    for k in range(nconv):
        errs[f"dW{k}"] = rel(dW[k], ws[k].grad.reshape(H, 16).t().reshape(-1))
        if bias:
            errs[f"db{k}"] = rel(db[k], bs[k].grad)
    return errs


@pytest.mark.parametrize("nconv,gln,mode,accumulate,bias", [
    (4, True, 0, False, False),   # the four global convolutions of the TFAR fusion layers (input G3)
    (2, True, 0, False, False),   # concat layer: global embedding + gate (input F1)
    (1, False, 1, False, False),  # fusion_layers[1].local_embedding (input gLN(D1))
    (1, False, 0, False, False),  # concat layer's local embedding (input F0)
    (1, False, 1, True, False),   # fusion_layers[0].local_embedding, accumulated into d(gLN(D0))
    (1, True, 2, False, True),    # downsample_layers[0]: gLN adjoint on load, input PReLU(gLN(y0)), bias
    (2, False, 2, True, True), (4, False, 1, True, False), (1, True, 0, True, False),
])
@pytest.mark.parametrize("B,T,Fq", [(2, 21, 19), (3, 8, 64), (1, 125, 64), (2, 33, 129), (1, 9, 8)])
def test_dw_adjoint_matches_float64_autograd(nconv, gln, mode, accumulate, bias, B, T, Fq):
    errs = _case(nconv, gln, mode, accumulate, bias, B, T, Fq, seed=7 * nconv + T)
    worst = max(errs.values())
    assert worst < 2e-5, errs  # fp32 sums over up to 8000 pixels x B against float64 (observed ~1e-6)


def test_dw_adjoint_refuses_unsupported_arguments():
    from rtfs_net_amd import lib

    t = torch.zeros(64 * 64 * 8, device="cuda")
    w = torch.zeros(1024, device="cuda")
    with pytest.raises(RuntimeError):
        lib.call("rtfs_dw_adjoint", 3, [t, t, t], None, None, None, None, [w, w, w], t, None, None, None, 0.0, 0, None, 0, 0, t.clone(), 0, [w.clone()] * 3, None, 1, 8, 64)
    with pytest.raises(RuntimeError):
        lib.call("rtfs_dw_adjoint", 1, [t], None, None, None, None, [w], t, None, None, None, 0.0, 1, None, 0, 0, t.clone(), 0, [w.clone()], None, 1, 8, 64)  # mode 1 without statistics


@pytest.mark.parametrize("mode,accumulate", [(0, False), (1, True)])
@pytest.mark.parametrize("B,T,Fq,Tg,Fg", [(2, 21, 19, 10, 9), (1, 33, 129, 16, 64), (2, 16, 64, 16, 64), (1, 251, 129, 125, 64)])
def test_dw_adjoint_mix_matches_float64_autograd(mode, accumulate, B, T, Fq, Tg, Fg):
    """rtfs_dw_adjoint_mix: the convolution is the LOCAL branch of an InjectionMultiSum (layers/fusion.py:54-69); the gradient handed over is the one of the
    mix's OUTPUT, the gate's sigmoid (nearest up-sampling Tg x Fg -> T x F, F.interpolate's floor(i * in / out)) and the local gLN adjoint are applied on load"""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(3 * T + Fq)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    x_in, taps = rnd(B, H, T, Fq), 0.3 * rnd(16, H)
    in_g, in_b = 1 + 0.3 * rnd(H), 0.2 * rnd(H)
    lg, lb, gg, gb = 1 + 0.3 * rnd(H), 0.1 * rnd(H), 1 + 0.3 * rnd(H), 0.1 * rnd(H)
    gate, dout = rnd(B, H, Tg, Fg), rnd(B, H, T, Fq)
    xin = (F.group_norm(x_in, 1, in_g, in_b, 1e-5) if mode else x_in).detach().requires_grad_(True)
    w = taps.t().reshape(H, 1, 4, 4).clone().requires_grad_(True)
    y = F.conv2d(F.pad(xin, (1, 2, 1, 2)), w, None, groups=H)
    s_up = F.interpolate(torch.sigmoid(F.group_norm(gate, 1, gg, gb, 1e-5)), size=(T, Fq), mode="nearest")
    (F.group_norm(y, 1, lg, lb, 1e-5) * s_up * dout).sum().backward()
    dev = "cuda"
    slot = lambda t: torch.stack([t.reshape(B, -1).sum(1), (t.reshape(B, -1) ** 2).sum(1)] + [torch.zeros(B, dtype=torch.float64)] * (lib.STAT_STRIDE - 2), 1).contiguous().to(dev)  # noqa: E731
    y32 = y.detach().float().double()
    mean = y32.reshape(B, -1).mean(1).view(B, 1, 1, 1)
    xh = (y32 - mean) / torch.sqrt((y32.reshape(B, -1) ** 2).mean(1).view(B, 1, 1, 1) - mean ** 2 + 1e-5)
    a = dout * s_up * lg.view(1, H, 1, 1)
    red = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64)
    red[:, 0], red[:, 1] = a.reshape(B, -1).sum(1), (a * xh).reshape(B, -1).sum(1)
    N = B * T * Fq * H
    dIn0 = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev) if accumulate else torch.full((N,), float("nan"), device=dev)
    dIn, dW = dIn0.clone(), torch.zeros(16 * H, device=dev)
    f32 = lambda t: t.float().contiguous().to(dev)  # noqa: E731
    sig = _cl(torch.sigmoid(F.group_norm(gate, 1, gg, gb, 1e-5))).to(dev)  # (what rtfs_mix_gln_bwd_sig's reduce pass writes)
    lib.call("rtfs_dw_adjoint_mix", _cl(dout).to(dev), _cl(y.detach()).to(dev), slot(y32), red.to(dev), f32(lg), sig, Tg, Fg, f32(taps), _cl(x_in).to(dev),
             slot(x_in.float().double()) if mode else None, f32(in_g) if mode else None, f32(in_b) if mode else None, 0.0, mode, None, 0, 0, dIn, 1 if accumulate else 0, dW, B, T, Fq)
    torch.cuda.synchronize()
    rel = lambda p, q: float((p.double().cpu() - q).norm() / q.norm())  # noqa: E731
    e_in = rel(dIn, _cl(xin.grad).double().reshape(-1) + (dIn0.double().cpu() if accumulate else 0))
    e_w = rel(dW, w.grad.reshape(H, 16).t().reshape(-1))
    assert e_in < 2e-5 and e_w < 2e-5, (e_in, e_w)


@pytest.mark.parametrize("nconv,B,T,Fq,Tg,Fg", [(1, 2, 21, 19, 10, 9), (2, 1, 125, 64, 125, 64), (2, 2, 16, 64, 16, 64), (1, 1, 251, 129, 125, 64)])
def test_dw_adjoint_with_a_tfar_mix_as_input(nconv, B, T, Fq, Tg, Fg):
    """input mode 3: the convolutions read the TFAR mix  gLN(l) * sigmoid(gLN(gate))^ + gLN(glob)^  (the concat layer reading the fusion layers' outputs,
    tdanet.py:124-129), which the kernel re-forms per pixel from the three tensors as rtfs_dwconv_mix_fwd does in the forward; gLN'd convolutions behind it"""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(11 * T + nconv)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    loc, gate, glob = rnd(B, H, T, Fq), rnd(B, H, Tg, Fg), rnd(B, H, Tg, Fg)
    aff = [(1 + 0.3 * rnd(H), 0.2 * rnd(H)) for _ in range(3)]
    up = lambda t: F.interpolate(t, size=(T, Fq), mode="nearest")  # noqa: E731
    xin = (F.group_norm(loc, 1, *aff[0], 1e-5) * up(torch.sigmoid(F.group_norm(gate, 1, *aff[1], 1e-5))) + up(F.group_norm(glob, 1, *aff[2], 1e-5))).detach().requires_grad_(True)
    taps = [0.3 * rnd(16, H) for _ in range(nconv)]
    gam, bet, dns = [1 + 0.3 * rnd(H) for _ in range(nconv)], [0.1 * rnd(H) for _ in range(nconv)], [rnd(B, H, T, Fq) for _ in range(nconv)]
    ws = [t.t().reshape(H, 1, 4, 4).clone().requires_grad_(True) for t in taps]
    ys, loss = [], 0
    for k in range(nconv):
        y = F.conv2d(F.pad(xin, (1, 2, 1, 2)), ws[k], None, groups=H)
        ys.append(y.detach())
        loss = loss + (F.group_norm(y, 1, gam[k], bet[k], 1e-5) * dns[k]).sum()
    loss.backward()
    dev = "cuda"
    slot = lambda t: torch.stack([t.reshape(B, -1).sum(1), (t.reshape(B, -1) ** 2).sum(1)] + [torch.zeros(B, dtype=torch.float64)] * (lib.STAT_STRIDE - 2), 1).contiguous().to(dev)  # noqa: E731
    f32 = lambda t: t.float().contiguous().to(dev)  # noqa: E731
    red = []
    for k in range(nconv):
        y32 = ys[k].float().double()
        mean = y32.reshape(B, -1).mean(1).view(B, 1, 1, 1)
        xh = (y32 - mean) / torch.sqrt((y32.reshape(B, -1) ** 2).mean(1).view(B, 1, 1, 1) - mean ** 2 + 1e-5)
        a = dns[k] * gam[k].view(1, H, 1, 1)
        r = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64)
        r[:, 0], r[:, 1] = a.reshape(B, -1).sum(1), (a * xh).reshape(B, -1).sum(1)
        red.append(r.to(dev))
    dIn = torch.full((B * T * Fq * H,), float("nan"), device=dev)
    dW = [torch.zeros(16 * H, device=dev) for _ in range(nconv)]
    in_mix = [_cl(gate).to(dev), slot(gate.float().double()), f32(aff[1][0]), f32(aff[1][1]), _cl(glob).to(dev), slot(glob.float().double()), f32(aff[2][0]), f32(aff[2][1])]
    lib.call("rtfs_dw_adjoint", nconv, [_cl(d).to(dev) for d in dns], [_cl(y).to(dev) for y in ys], [slot(y.float().double()) for y in ys], red, [f32(t) for t in gam],
             [f32(t) for t in taps], _cl(loc).to(dev), slot(loc.float().double()), f32(aff[0][0]), f32(aff[0][1]), 0.0, 3, in_mix, Tg, Fg, dIn, 0, dW, None, B, T, Fq)
    torch.cuda.synchronize()
    rel = lambda p, q: float((p.double().cpu() - q).norm() / q.norm())  # noqa: E731
    errs = [rel(dIn, _cl(xin.grad).double().reshape(-1))] + [rel(dW[k], ws[k].grad.reshape(H, 16).t().reshape(-1)) for k in range(nconv)]
    assert max(errs) < 2e-5, errs


@pytest.mark.parametrize("n", [1, 2, 5, 8])
def test_sum_n_is_the_left_to_right_fp32_sum(n):
    """rtfs_sum_n (csrc/bwd_misc.hip): out = x[0] + x[1] + ... in one launch - the gradient of the audio embedding a0, which every RTFS block after the
    first adds to its input (TDAVNet.py:109-113's residual), summed once instead of read-modify-written by each block's last kernel.  Bit-exact against the
    same left-to-right fp32 sum; refuses more than 8 terms, a count that is no multiple of 4, and a null term."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(n)
    count = 4 * 70001
    xs = [torch.randn(count, generator=g).cuda() for _ in range(n)]
    out = torch.full((count,), float("nan"), device="cuda")
    lib.call("rtfs_sum_n", xs, n, out, count)
    want = xs[0].clone()
    for x in xs[1:]:
        want += x
    assert torch.equal(out, want)
    with pytest.raises(RuntimeError):
        lib.call("rtfs_sum_n", xs, 9, out, count)
    with pytest.raises(RuntimeError):
        lib.call("rtfs_sum_n", xs, n, out, count - 1)


def _gemm_adjoint_inputs(B, rows, seed):
    g = torch.Generator().manual_seed(seed)
    dz = torch.randn(B, rows, 256, generator=g).cuda()
    x = torch.randn(B, rows, 256, generator=g).cuda()
    Wt = (torch.randn(256, 256, generator=g) / 16).cuda()  # [out of the adjoint = in of the forward][k]: Y = X . Wt^T as rtfs_gemm_rows takes it
    return dz, x, Wt


@pytest.mark.parametrize("B,rows", [(2, 777), (32, 8200)])
def test_gemm_prelu_bwd_against_the_two_launches_and_float64(B, rows):
    """rtfs_gemm_prelu_bwd: dx = prelu'(x) * (dz . Wt^T) and dslope in one launch (csrc/gemm.hip: ws256_kernel<ProPlain, EpiAdjoint<1>>, the weight-stationary
    form, at the large map; rtfs_gemm_rows + rtfs_prelu_bwd in place at the small one) - the adjoint of mask_generator.py:47-48's PReLU -> Conv2d w.r.t. the
    PReLU's input.  dx bit-identical to the two launches it replaces (same products in the same order), dslope against float64."""
    from rtfs_net_amd import lib

    dz, x, Wt = _gemm_adjoint_inputs(B, rows, 5)
    slope = 0.25
    dx, dsl = torch.full_like(x, float("nan")), torch.zeros(1, device="cuda")
    lib.call("rtfs_gemm_prelu_bwd", dz, Wt, x, slope, dx, dsl, B, rows)
    dpre, dx2, dsl2 = torch.empty_like(x), torch.empty_like(x), torch.zeros(1, device="cuda")
    lib.call("rtfs_gemm_rows", dz, Wt, None, dpre, B * rows, 256, 256, 0)
    lib.call("rtfs_prelu_bwd", dpre, x, slope, dx2, 0, dsl2, B * rows * 256)
    assert torch.equal(dx, dx2)
    pre64 = dz.double() @ Wt.double().t()
    want = float((pre64 * x.double() * (x <= 0)).sum())
    scale = float((pre64 * x.double() * (x <= 0)).abs().sum()) ** 0.5 + 1.0
    assert abs(float(dsl) - want) <= 2e-3 * scale, (float(dsl), want)
    assert abs(float(dsl2) - want) <= 2e-3 * scale
    # accumulates into dslope
    lib.call("rtfs_gemm_prelu_bwd", dz, Wt, x, slope, dx, dsl, B, rows)
    assert abs(float(dsl) - 2 * want) <= 4e-3 * scale


@pytest.mark.parametrize("B,rows", [(2, 777), (32, 8200)])
def test_gemm_gln_relu_bwd_reduce_against_the_two_launches(B, rows):
    """rtfs_gemm_gln_relu_bwd_reduce: dR = dy . Wt^T with the reduce pass of relu(gLN(x))'s adjoint (S1, S2 per utterance, dgamma, dbeta) in the GEMM's epilogue
    (tdavnet.py:59,89: audio_bottleneck = pre-norm gLN + pre-act ReLU + Conv2d) against rtfs_gemm_rows + rtfs_gln_bwd_reduce(act 2): dR bit-identical, the sums to
    fp32 summation-order tolerance, and against float64."""
    from rtfs_net_amd import lib

    dy, x, Wt = _gemm_adjoint_inputs(B, rows, 9)
    g = torch.Generator().manual_seed(1)
    gamma, beta = (1 + 0.3 * torch.randn(256, generator=g)).cuda(), (0.2 * torch.randn(256, generator=g)).cuda()
    st = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
    st[:, 0], st[:, 1] = x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))
    out = []
    for fused in (True, False):
        dR = torch.full_like(x, float("nan"))
        red = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
        dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
        if fused:
            lib.call("rtfs_gemm_gln_relu_bwd_reduce", dy, Wt, x, st, gamma, beta, dR, red, dg, db, B, rows)
        else:
            lib.call("rtfs_gemm_rows", dy, Wt, None, dR, B * rows, 256, 256, 0)
            lib.call("rtfs_gln_bwd_reduce", dR, x, st, gamma, beta, 2, 0.0, red, dg, db, None, B, rows, 256)
        out.append((dR, red[:, :2].clone(), dg, db))
    assert torch.equal(out[0][0], out[1][0])
    # float64 restatement
    n = rows * 256
    mean = (st[:, 0] / n).view(B, 1, 1)
    rstd = 1.0 / torch.sqrt((st[:, 1] / n).view(B, 1, 1) - mean ** 2 + 1e-5)
    xh = (x.double() - mean) * rstd
    gg = (dy.double() @ Wt.double().t()) * ((xh * gamma.double() + beta.double()) > 0)
    want = (torch.stack([(gg * gamma.double()).sum((1, 2)), (gg * gamma.double() * xh).sum((1, 2))], 1), (gg * xh).sum((0, 1)), gg.sum((0, 1)))
    for got in out:
        for a, b_ in zip(got[1:], want):
            assert float((a.double() - b_).abs().max()) <= 2e-4 * float(b_.abs().max()) + 1e-3 * (rows * B) ** 0.5, (a, b_)


@pytest.mark.parametrize("rows", [64 * 16 * 3 + 37, 100003])
def test_proj_gateway_bwd_next_equals_the_two_launches(rows):
    """rtfs_proj_gateway_bwd_next: the gateway / projection adjoint (tdanet.py:108-109) + the NEXT block's residual-conv input gradient dE = ds . Wr^T (tdanet.py:127-131)
    from the ds rows while they are in LDS, against rtfs_proj_gateway_bwd followed by rtfs_gemm_rows(ds, WrT, 256 -> 64) on the stored ds: ds bit-identical, dE the same
    products (observed bit-identical; held to 1e-6), parameter sums to fp32 atomics' order; ragged last tile, guard rows behind dE."""
    from rtfs_net_amd import lib

    g = torch.Generator().manual_seed(rows)
    dy0, dx, s = torch.randn(rows, 64, generator=g).cuda(), torch.randn(rows, 256, generator=g).cuda(), torch.randn(rows, 256, generator=g).cuda()
    WpT, WrT = (torch.randn(256, 64, generator=g) / 8).cuda(), (torch.randn(64, 256, generator=g) / 16).cuda()
    gw, gb = (1 + 0.3 * torch.randn(256, generator=g)).cuda(), (0.2 * torch.randn(256, generator=g)).cuda()
    out = []
    for fused in (True, False):
        ds = torch.full((rows, 256), float("nan"), device="cuda")
        dE = torch.cat([torch.full((rows, 64), float("nan")), torch.full((64, 64), 7.0)]).cuda()
        dgw, dgb, dsl = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda"), torch.zeros(1, device="cuda")
        if fused:
            lib.call("rtfs_proj_gateway_bwd_next", dy0, WpT, dx, s, gw, gb, 0.25, ds, dgw, dgb, dsl, WrT, dE, rows)
        else:
            lib.call("rtfs_proj_gateway_bwd", dy0, WpT, dx, s, gw, gb, 0.25, ds, 0, None, 0, dgw, dgb, dsl, rows)
            lib.call("rtfs_gemm_rows", ds, WrT, None, dE, rows, 256, 64, 0)
        out.append((ds, dE, dgw, dgb, dsl))
    a, b = out
    assert torch.equal(a[0], b[0])
    assert bool((a[1][rows:] == 7.0).all()) and bool(torch.isfinite(a[1][:rows]).all())
    assert float((a[1][:rows] - b[1][:rows]).norm()) <= 1e-6 * float(b[1][:rows].norm())
    for x, y in zip(a[2:], b[2:]):
        assert float((x - y).norm()) <= 1e-4 * float(y.norm()) + 1e-6
