"""GPU micro-benchmark of the 256 -> 256 pixel GEMM entry points (audio bottleneck, S3 mask, plain rows GEMM of the training step) at the
bench shape, with output checksums for same-box A/B runs and a float64 check of a row sample.

    [RTFS_HIP_LIB=exp/old/librtfs_hip.so] python tools/pix_bench.py [B] [T]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def timeit(fn, n=30):
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
    return 1e3 * t[len(t) // 2], 1e3 * t[0]


def main(B=32, T=251):
    TF = T * 129
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, TF, 256, generator=g).cuda()
    emb = torch.randn(B, TF, 256, generator=g).cuda()
    W = (torch.randn(256, 256, generator=g) * 0.06).cuda()
    bias = (torch.randn(256, generator=g) * 0.1).cuda()
    gamma, beta = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
    stats = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
    stats[:, 0] = x.double().sum((1, 2))
    stats[:, 1] = (x.double() ** 2).sum((1, 2))
    y = torch.empty_like(x)
    fl = 2.0 * B * TF * 256 * 256
    rows = torch.tensor([0, 1, 63, 64, TF - 1, TF - 2, TF // 2, 12345 % TF])

    def report(name, fn, ref_fn):
        med, mn = timeit(fn)
        yb = y.view(B, TF, 256)
        err = 0.0
        for b in (0, B - 1):
            ref = ref_fn(b)
            err = max(err, float((yb[b, rows].double() - ref).abs().max() / ref.abs().max()))
        print(f"{name}: median {med:.1f} us  min {mn:.1f} us  {fl / (med * 1e-6) / 1e12:.1f} TFLOP/s = {fl / (med * 1e-6) / 157.3e12:.3f} of the fp32 MFMA peak"
              f"   checksum {float(y.double().sum()):.12e} {float(y.double().abs().sum()):.12e}   max rel err vs float64 (row sample) {err:.2e}", flush=True)

    Wd = W.double()

    def ref_bn(b):
        n = TF * 256
        mean = stats[b, 0] / n
        var = stats[b, 1] / n - mean * mean
        xn = (x[b, rows].double() - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
        return torch.relu(xn) @ Wd.t() + bias.double()

    report("rtfs_bottleneck_fwd", lambda: lib.call("rtfs_bottleneck_fwd", x, stats, gamma, beta, W, bias, y, B, TF), ref_bn)

    def ref_mask(b):
        xr = x[b, rows].double()
        m = torch.relu(torch.where(xr >= 0, xr, 0.25 * xr) @ Wd.t() + bias.double())
        mr, mi = m[:, :128], m[:, 128:]
        er, ei = emb[b, rows, :128].double(), emb[b, rows, 128:].double()
        return torch.cat([er * mr - ei * mi, er * mi + ei * mr], 1)

    report("rtfs_mask_fwd", lambda: lib.call("rtfs_mask_fwd", x, 0.25, W, bias, emb, y, None, B, TF), ref_mask)
    xf = x.view(B * TF, 256)
    report("rtfs_gemm_rows (256 -> 256, no bias)", lambda: lib.call("rtfs_gemm_rows", xf, W, None, y, B * TF, 256, 256, 0),
           lambda b: x[b, rows].double() @ Wd.t())


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
