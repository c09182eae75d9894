"""Same-box timing of rtfs_dw_adjoint (one launch) against the launches it replaces (rtfs_gln_bwd_apply + rtfs_dwconv_bwd_weight + rtfs_dwconv_bwd_input per
convolution) for the six convolution groups of an RTFS block's backward at the config-3 shape (32 utterances: full resolution 251 x 129, compressed 125 x 64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib  # noqa: E402

B = int(os.environ.get("B", "32"))
dev = "cuda"
GROUPS = [("global x4 (G3)", 4, True, 0, False, False, 125, 64), ("concat g+gate x2 (F1)", 2, True, 0, False, False, 125, 64),
          ("f1l (gLN(D1))", 1, False, 1, False, False, 125, 64), ("cl (F0)", 1, False, 0, False, False, 251, 129),
          ("f0l (gLN(D0)) +=", 1, False, 1, True, False, 251, 129), ("d0 (PReLU(gLN(y0)))", 1, True, 2, False, True, 251, 129)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return 1e3 * t[len(t) // 2]


tot_old = tot_new = 0.0
for name, nconv, gln, mode, acc, bias, T, Fq in GROUPS:
    N = B * T * Fq * 64
    r = lambda: torch.randn(N, device=dev)  # noqa: E731
    dy, x = [r() for _ in range(nconv)], [r() for _ in range(nconv)]
    st = [torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device=dev) for _ in range(nconv + 1)]
    for s in st:
        s[:, 0], s[:, 1] = 0.0, float(T * Fq * 64)
    red = [torch.randn(B, lib.STAT_STRIDE, dtype=torch.float64, device=dev) for _ in range(nconv)]
    gam = [torch.randn(64, device=dev) for _ in range(nconv + 1)]
    bet = [torch.randn(64, device=dev) for _ in range(nconv + 1)]
    w = [torch.randn(1024, device=dev) for _ in range(nconv)]
    xin, dIn = r(), r()
    dW, db = [torch.zeros(1024, device=dev) for _ in range(nconv)], [torch.zeros(64, device=dev) for _ in range(nconv)]
    dX = [r() for _ in range(nconv)]

    def old():
        for k in range(nconv):
            src = dy[k]
            if gln:
                lib.call("rtfs_gln_bwd_apply", dy[k], x[k], st[k], gam[k], bet[k], 0, 0.0, red[k], dX[k], 0, B, T * Fq, 64)
                src = dX[k]
            lib.call("rtfs_dwconv_bwd_weight", src, xin, st[nconv] if mode else None, gam[nconv] if mode else None, bet[nconv] if mode else None, 0.25, mode, 1, dW[k],
                     db[k] if bias else None, B, T, Fq)
            lib.call("rtfs_dwconv_bwd_input", src, w[k], dIn, 1 if (acc or k > 0) else 0, 1, B, T, Fq)

    def new():
        lib.call("rtfs_dw_adjoint", nconv, dy, x if gln else None, st[:nconv] if gln else None, red if gln else None, gam[:nconv] if gln else None, w, xin,
                 st[nconv] if mode else None, gam[nconv] if mode else None, bet[nconv] if mode else None, 0.25, mode, None, 0, 0, dIn, 1 if acc else 0, dW, db if bias else None, B, T, Fq)

    t_old, t_new = timeit(old), timeit(new)
    units_new = nconv * (2 if gln else 1) + 1 + 1 + (1 if acc else 0)
    gb = units_new * N * 4 / 1e9
    tot_old, tot_new = tot_old + t_old, tot_new + t_new
    print(f"{name:28s} {t_old:8.1f} us -> {t_new:8.1f} us   ({gb:.2f} GB algorithmic: {gb / t_new * 1e3:.2f} TB/s)", flush=True)
print(f"per block: {tot_old:.0f} -> {tot_new:.0f} us")

# the local branches of the three mixes: rtfs_mix_gln_bwd's apply pass + the plain adjoint, against rtfs_dw_adjoint_mix
for name, mode, acc, T, Fq, Tg, Fg in (("cl <- mix", 0, False, 251, 129, 125, 64), ("f0l <- mix +=", 1, True, 251, 129, 125, 64), ("f1l <- mix", 1, False, 125, 64, 125, 64)):
    N, Ng = B * T * Fq * 64, B * Tg * Fg * 64
    r = lambda n=N: torch.randn(n, device=dev)  # noqa: E731
    dOut, loc, xin, dIn, dLoc = r(), r(), r(), r(), r()
    gate, sig = r(Ng), torch.rand(Ng, device=dev)
    st = [torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device=dev) for _ in range(3)]
    for s_ in st:
        s_[:, 1] = float(T * Fq * 64)
    red = torch.randn(B, lib.STAT_STRIDE, dtype=torch.float64, device=dev)
    gam, bet, w, dW = torch.randn(64, device=dev), torch.randn(64, device=dev), torch.randn(1024, device=dev), torch.zeros(1024, device=dev)
    m = (st[2], gam, bet) if mode else (None, None, None)

    def old():
        # (the apply pass alone: same traffic as mix_gln_bwd_apply_kernel - dOut, loc, gate in, dLoc out)
        lib.call("rtfs_gln_bwd_apply", dOut, loc, st[0], gam, bet, 0, 0.0, red, dLoc, 0, B, T * Fq, 64)
        lib.call("rtfs_dw_adjoint", 1, [dLoc], None, None, None, None, [w], xin, *m, 0.0, mode, None, 0, 0, dIn, 1 if acc else 0, [dW], None, B, T, Fq)

    def new():
        lib.call("rtfs_dw_adjoint_mix", dOut, loc, st[0], red, gam, sig, Tg, Fg, w, xin, *m, 0.0, mode, None, 0, 0, dIn, 1 if acc else 0, dW, B, T, Fq)

    t_old, t_new = timeit(old), timeit(new)
    gb = (4 + (1 if acc else 0)) * N * 4 / 1e9 + Ng * 4 / 1e9
    print(f"{name:28s} {t_old:8.1f} us -> {t_new:8.1f} us   ({gb:.2f} GB algorithmic: {gb / t_new * 1e3:.2f} TB/s)", flush=True)
