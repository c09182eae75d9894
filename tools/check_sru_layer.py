"""GPU check: rtfs_sru_layer_fwd == rtfs_gemm_rows_fwd(64->192) + rtfs_sru_scan_fwd(km=3) on random data (ragged lengths)."""
import sys

import torch

sys.path.insert(0, "/root/repo")
from rtfs_net_amd import lib  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
for S, L in ((5, 57), (3, 118), (2, 9), (4, 32), (1, 33), (7, 1)):
    h = torch.randn(S * L * 64, device=dev)
    W = torch.randn(192, 64, device=dev) * 0.2
    wc, bias = torch.randn(128, device=dev) * 0.5, torch.randn(128, device=dev) * 0.5
    U = torch.empty(S * L * 192, device=dev)
    lib.call("rtfs_gemm_rows_fwd", h, W, None, U, S * L, 64, 192)
    ref = torch.empty_like(h)
    lib.call("rtfs_sru_scan_fwd", U, h, wc, bias, 0.7, ref, S, L, 3)
    out, cst = torch.empty_like(h), torch.empty_like(h)
    lib.call("rtfs_sru_layer_fwd", h, W, wc, bias, 0.7, out, None, None, S, L)
    out2 = torch.empty_like(h)
    U2 = torch.empty_like(U)
    lib.call("rtfs_sru_layer_fwd", h, W, wc, bias, 0.7, out2, cst, U2, S, L)
    torch.cuda.synchronize()
    print(S, L, float((out - ref).abs().max()), float((out2 - ref).abs().max()), "U", float((U2 - U).abs().max()))
