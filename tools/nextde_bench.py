"""Same-box timing of rtfs_proj_gateway_bwd_next (one launch) against rtfs_proj_gateway_bwd + rtfs_gemm_rows(256 -> 64) at the config-3 shape (32 x 251 x 129 rows)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib  # noqa: E402

rows = int(os.environ.get("B", "32")) * 251 * 129
dy0, dx, s = torch.randn(rows, 64, device="cuda"), torch.randn(rows, 256, device="cuda"), torch.randn(rows, 256, device="cuda")
WpT, WrT = torch.randn(256, 64, device="cuda") / 8, torch.randn(64, 256, device="cuda") / 16
gw, gb = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.2
ds, dE = torch.empty(rows, 256, device="cuda"), torch.empty(rows, 64, device="cuda")
dgw, dgb, dsl = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda"), torch.zeros(1, device="cuda")


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


print("rtfs_proj_gateway_bwd        %8.1f us" % timeit(lambda: lib.call("rtfs_proj_gateway_bwd", dy0, WpT, dx, s, gw, gb, 0.25, ds, 0, None, 0, dgw, dgb, dsl, rows)))
print("rtfs_gemm_rows 256 -> 64     %8.1f us" % timeit(lambda: lib.call("rtfs_gemm_rows", ds, WrT, None, dE, rows, 256, 64, 0)))
print("rtfs_proj_gateway_bwd_next   %8.1f us" % timeit(lambda: lib.call("rtfs_proj_gateway_bwd_next", dy0, WpT, dx, s, gw, gb, 0.25, ds, dgw, dgb, dsl, WrT, dE, rows)))
