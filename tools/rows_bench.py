"""rtfs_gemm_rows at the training step's narrow shapes (N = 64): time, achieved HBM rate, error against float64.   python tools/rows_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib
g = torch.Generator().manual_seed(5)
for M, K, acc in ((4000 * 57, 192, 1), (2048 * 118, 192, 1), (32 * 251 * 129, 256, 0), (40000, 192, 1), (256000, 96, 1), (256000, 64, 0)):
    X = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(64, K, generator=g) * 0.1).cuda()
    Y0 = torch.randn(M, 64, generator=g).cuda()
    Y = Y0.clone()
    lib.call("rtfs_gemm_rows", X, W, None, Y, M, K, 64, acc)
    want = X[:100000].double() @ W.double().t() + (Y0[:100000].double() if acc else 0)
    err = float((Y[:100000].double() - want).norm() / want.norm())
    tail = X[-1000:].double() @ W.double().t() + (Y0[-1000:].double() if acc else 0)
    err_t = float((Y[-1000:].double() - tail).norm() / tail.norm())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(); lib.call("rtfs_gemm_rows", X, W, None, Y, M, K, 64, acc); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[15]
    by = 4.0 * M * (K + 64 * (2 if acc else 1))
    print(f"M {M} K {K} accumulate {acc}: {1e3 * t:.1f} us  {by / (t * 1e-3) / 1e12:.2f} TB/s  rel err vs float64: first 100k rows {err:.1e}, last 1000 rows {err_t:.1e}")
