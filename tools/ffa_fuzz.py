"""Random-shape fuzz of the three fast-FIR entry points (layer-0 unfold GEMM - also in the six-term split -, ConvTranspose1d forward, its input gradient) against float64 on the GPU:
every output ROW on its own, at shapes that take the fast-FIR kernels (pair rows per sequence Lv from 21 up, two- to four-sequence tiles, ragged last tile).
    python tools/ffa_fuzz.py [n_shapes] [seed]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def rows_err(got, want):
    d = (got.double() - want).norm(dim=-1) / want.norm(dim=-1).clamp_min(1e-30)
    return float(d.max()), int(d.argmax())


def one(B, T2, dim, g):
    G = torch.randn(B, T2, 64, 64, generator=g).cuda()
    gamma, beta = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    W0 = (torch.randn(256, 512, generator=g) * 0.05).cuda()
    Wc = (torch.randn(64, 512, generator=g) * 0.05).cuda()
    bias = (torch.randn(64, generator=g) * 0.1).cuda()
    S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
    L = npos - 7
    out = []
    # layer-0 GEMM
    x = G.double()
    xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    seqs = xn.reshape(S, 64, 64) if dim == 4 else xn.permute(0, 2, 1, 3).reshape(S, T2, 64)
    want = torch.zeros(S, L, 256, dtype=torch.float64, device="cuda")
    for k in range(8):
        want += seqs[:, k:k + L] @ W0.double()[:, 64 * k:64 * k + 64].t()
    U = torch.full((S * L * 256,), float("nan"), device="cuda")
    lib.call("rtfs_dp_unfold_gemm_fwd", G, gamma, beta, W0, U, B, T2, dim, 0)
    out.append(("unfold", rows_err(U.view(S, L, 256), want), (S * ((L + 2) // 2) + 62) // 63 >= 512 and (L + 2) // 2 >= 21))
    # the same GEMM in the six-term split (unfold_ws6_kernel from 512 flattened 64-row tiles on, L >= 32)
    U6 = torch.full((S * L * 256,), float("nan"), device="cuda")
    lib.call("rtfs_dp_unfold_gemm_fwd_bf16", G, gamma, beta, W0, U6, B, T2, dim, 0, 6)
    out.append(("unfold_x6", rows_err(U6.view(S, L, 256), want), (S * L + 63) // 64 >= 512 and L >= 32))
    # ConvTranspose forward (in place on a copy of G)
    H3 = torch.randn(S, L, 64, generator=g).cuda()
    hp = torch.zeros(S, npos + 14, 64, dtype=torch.float64, device="cuda")
    hp[:, 7:7 + L] = H3.double()
    y = torch.zeros(S, npos, 64, dtype=torch.float64, device="cuda")
    for k in range(8):
        y += hp[:, k:k + npos] @ Wc.double()[:, 64 * k:64 * k + 64].t()
    y += bias.double()
    G0s = (G if dim == 4 else G.permute(0, 2, 1, 3)).reshape(S, npos, 64).double()
    Gc = G.clone()
    lib.call("rtfs_dp_convt_fwd", H3, Wc, bias, Gc, B, T2, dim)
    got = (Gc if dim == 4 else Gc.permute(0, 2, 1, 3)).reshape(S, npos, 64)
    out.append(("convt", rows_err(got, y + G0s), (S * ((npos + 2) // 2) + 62) // 63 >= 1024 and (npos + 2) // 2 >= 21))
    # ConvTranspose input gradient
    seqg = (G if dim == 4 else G.permute(0, 2, 1, 3)).reshape(S, npos, 64).double()
    wantb = torch.zeros(S, L, 64, dtype=torch.float64, device="cuda")
    for k in range(8):
        wantb += seqg[:, k:k + L] @ Wc.double()[:, 64 * k:64 * k + 64].t()
    dH = torch.full((S * L * 64,), float("nan"), device="cuda")
    lib.call("rtfs_convt_bwd_input", G, Wc, dH, B, T2, dim)
    out.append(("convt_bwd", rows_err(dH.view(S, L, 64), wantb), (S * ((L + 2) // 2) + 62) // 63 >= 1024 and (L + 2) // 2 >= 21))
    return out


def main(n=40, seed=0):
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    worst = {}
    for it in range(n):
        dim = rnd.choice((3, 4))
        if dim == 3:  # sequences along T: Lv = (T2 - 5) // 2 from 21 up
            T2 = rnd.choice((47, 48, 49, 50, 51, 63, 64, 77, 100, 125, 126, 201, 250))
            lo = max(1, 70000 // (64 * max(1, (T2 - 5) // 2)))
            B = rnd.randint(lo, max(lo + 3, 40))
        else:
            T2 = rnd.choice((16, 31, 40, 125, 250))
            B = rnd.randint(max(1, 2300 // T2 + 1), max(2300 // T2 + 2, 4800 // T2))
        if B * T2 * 64 * 64 * 4 * 4 > 6e9:
            continue
        res = one(B, T2, dim, g)
        line = f"B {B:3d} T2 {T2:3d} dim {dim}: " + "  ".join(f"{nm} {e:.2e}{'*' if ffa else ' '}" for nm, (e, _), ffa in res)
        print(line, flush=True)
        for nm, (e, row), ffa in res:
            if e > worst.get(nm, (0,))[0]:
                worst[nm] = (e, B, T2, dim, row)
            assert e < 5e-6, (nm, B, T2, dim, e, row)
    print("worst single-row relative error per entry point (* = fast-FIR kernel):", worst)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
