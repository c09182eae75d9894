"""Time the adjoint of one SRU layer 1-3 at the training shapes: the three launches (rtfs_sru_scan_bwd + rtfs_wgrad + rtfs_gemm_rows) against the
one-launch form (rtfs_sru_layer_bwd).  Usage: python tools/sru_bwd_bench.py [B]"""
import sys

import torch

sys.path.insert(0, ".")
from rtfs_net_amd import lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for name, S, L in (("freq", B * 125, 57), ("time", B * 64, 118)):
    g = torch.Generator().manual_seed(1)
    X = torch.randn(S, L, 64, generator=g).cuda()
    W = (torch.randn(192, 64, generator=g) * 0.15).cuda()
    Wt = W.t().contiguous()
    wc, bias = (torch.randn(128, generator=g) * 0.5).cuda(), (torch.randn(128, generator=g) * 0.5).cuda()
    dH = torch.randn(S, L, 64, generator=g).cuda()
    H, C, U = torch.empty_like(X), torch.empty_like(X), torch.empty(S, L, 192, device="cuda")
    lib.call("rtfs_sru_layer_fwd", X, W, wc, bias, 1.7, H, C, U, S, L)
    dU, dX, dXb = torch.empty_like(U), torch.empty_like(X), torch.empty_like(X)
    dwc, db, dW = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda"), torch.zeros(192 * 64, device="cuda")

    work = torch.empty(lib.load().rtfs_sru_layer_bwd_work_floats(S), device="cuda")

    def three():
        lib.call("rtfs_sru_scan_bwd", U, X, C, wc, bias, 1.7, dH, dU, dX, dwc, db, S, L, 3)
        lib.call("rtfs_wgrad", dU, 192, X, 64, dW, 64, None, S * L, 0, 0, 0, 1, 192, 64, 0, None, None, 0.0, None, 0)
        lib.call("rtfs_gemm_rows", dU, Wt, None, dX, S * L, 192, 64, 1)

    def one():
        lib.call("rtfs_sru_layer_bwd", U, X, C, W, wc, bias, 1.7, dH, dH, dX, dXb, work, dW, dwc, db, S, L)

    for label, fn in (("three launches", three), ("one launch", one), ("three launches", three), ("one launch", one)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name} S={S} L={L} {label}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    if "LB_TIMING" in __import__("os").environ:  # the library was built with -DLB_TIMING: per-wave s_memtime ticks (100 per us) per phase sit in dX
        one()
        torch.cuda.synchronize()
        nw = min((S // 2 + 3) // 4, 256) * 8
        t = dX.view(-1).view(torch.int64)[: nw * 8].view(nw, 8).double().cpu()
        names = ("chunk end -> loop top (first: weight staging, first loads)", "dW operand loads + recurrence (incl. waits for its operands)", "issue next loads", "dW MFMAs", "dX MFMAs", "stage + store", "-", "tail: dW reduction")
        tot = t.sum(dim=1)
        print(f"  per-wave ticks: total median {tot.median():.0f} max {tot.max():.0f}")
        for k, nm in enumerate(names):
            print(f"    {nm}: median {t[:, k].median():.0f} ({100 * t[:, k].median() / tot.median():.0f} %)")
