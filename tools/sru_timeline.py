"""Phase timeline of sru_layer_kernel from a -DSRU_TIMING build (tools/build_variant.sh srutime -DSRU_TIMING; RTFS_HIP_LIB=exp/srutime/librtfs_hip.so):
per wave s_memtime stamps at kernel entry, and per chunk at (A fragments arrived, MFMAs + half exchange done, recurrence done)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402

B = 32
g = torch.Generator().manual_seed(0)
W = (torch.randn(192, 64, generator=g) * 0.1).cuda()
wc = (torch.rand(128, generator=g) * 2 - 1).cuda()
bias = (torch.randn(128, generator=g) * 0.1).cuda()
for S, L in ((B * 125, 57), (B * 64, 118)):
    h = torch.randn(S, L, 64, generator=g).cuda()
    out = torch.empty_like(h)
    for _ in range(3):
        lib.call("rtfs_sru_layer_fwd", h, W, wc, bias, 1.0, out, None, None, S, L)
    torch.cuda.synchronize()
    t = out.view(torch.int64).view(S, -1)[:, :24].cpu()
    nch = (L + 31) // 32
    t0 = t[:, 0].min()
    rel = (t[:, :1 + 5 * nch] - t0).double()
    names = ["entry"] + [f"ch{c}.{p}" for c in range(nch) for p in ("A-ready", "mfma-done", "barrier1", "scan-end", "barrier2")]
    print(f"S {S} L {L}: per-wave stamps relative to the first wave's entry (cycles of s_memtime): median / p10 / p90 / max")
    for k, n in enumerate(names):
        col = rel[:, k]
        print(f"  {n:16s} {col.median():10.0f} {col.quantile(0.1):10.0f} {col.quantile(0.9):10.0f} {col.max():10.0f}")
    d = rel[:, 1:].diff(dim=1)
    print("  phase durations (median): " + ", ".join(f"{names[k + 2]}-{names[k + 1]} {d[:, k].median():.0f}" for k in range(d.shape[1])))
    print(f"  total span {rel.max():.0f} ticks")
