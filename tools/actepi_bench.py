"""Same-box timing of rtfs_gemm_prelu_bwd / rtfs_gemm_gln_relu_bwd_reduce (one launch each) against the launches they replace, at the config-3 shape
(32 utterances x 251 x 129 pixels x 256 channels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib  # noqa: E402

B, rows = int(os.environ.get("B", "32")), 251 * 129


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


dz, x = torch.randn(B, rows, 256, device="cuda"), torch.randn(B, rows, 256, device="cuda")
Wt = torch.randn(256, 256, device="cuda") / 16
dx, tmp, dsl = torch.empty_like(x), torch.empty_like(x), torch.zeros(1, device="cuda")
gamma, beta = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.2
st = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
st[:, 1] = float(rows * 256)
red = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")

print("gemm alone                     %8.1f us" % timeit(lambda: lib.call("rtfs_gemm_rows", dz, Wt, None, tmp, B * rows, 256, 256, 0)))
print("prelu_bwd alone                %8.1f us" % timeit(lambda: lib.call("rtfs_prelu_bwd", tmp, x, 0.25, dx, 0, dsl, B * rows * 256)))
print("rtfs_gemm_prelu_bwd            %8.1f us" % timeit(lambda: lib.call("rtfs_gemm_prelu_bwd", dz, Wt, x, 0.25, dx, dsl, B, rows)))
print("gln_bwd_reduce alone           %8.1f us" % timeit(lambda: lib.call("rtfs_gln_bwd_reduce", tmp, x, st, gamma, beta, 2, 0.0, red, dg, db, None, B, rows, 256)))
print("rtfs_gemm_gln_relu_bwd_reduce  %8.1f us" % timeit(lambda: lib.call("rtfs_gemm_gln_relu_bwd_reduce", dz, Wt, x, st, gamma, beta, tmp, red, dg, db, B, rows)))
