"""GPU micro-benchmark of the weight-gradient GEMM entry point (rtfs_wgrad) at the shapes of the training step (B = 32, 2 s): plain maps
(residual conv 64 -> 256, gateway + projection 256 -> 64, 256 -> 256, SRU layers 64 -> 192) and the layer-0 Toeplitz form; checks a float64 sample."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def timeit(fn, n=12):
    for _ in range(2):
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


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    R = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    M = 32 * 251 * 129
    print(lib.library_path())
    for name, NOUT, KIN, pro, rows in (("resid conv dW (64 -> 256)", 256, 64, 0, M), ("gateway + projection dW (256 -> 64)", 64, 256, 1, M), ("256 -> 256 prelu", 256, 256, 2, M),
                                       ("SRU layer dW (64 -> 192), freq", 192, 64, 0, 4000 * 57), ("SRU layer dW (64 -> 192), time", 192, 64, 0, 2048 * 118)):
        dY, X = R(rows, NOUT), R(rows, KIN)
        p0, p1 = R(KIN) * 0.2 + 1, R(KIN) * 0.1
        dW, db = torch.zeros(NOUT, KIN, device="cuda"), torch.zeros(NOUT, device="cuda")
        t = timeit(lambda: lib.call("rtfs_wgrad", dY, NOUT, X, KIN, dW, KIN, db, rows, 0, 0, 0, 1, NOUT, KIN, pro, p0, p1, 0.25, None, 0))
        dW.zero_()
        lib.call("rtfs_wgrad", dY, NOUT, X, KIN, dW, KIN, db, rows, 0, 0, 0, 1, NOUT, KIN, pro, p0, p1, 0.25, None, 0)
        Xd = X[:200000].double()
        if pro == 1:
            u = Xd * p0.double() + p1.double()
            Xd = torch.where(u >= 0, u, 0.25 * u)
        if pro == 2:
            Xd = torch.where(Xd >= 0, Xd, 0.25 * Xd)
        dW2 = torch.zeros(NOUT, KIN, device="cuda")
        lib.call("rtfs_wgrad", dY, NOUT, X, KIN, dW2, KIN, None, 200000, 0, 0, 0, 1, NOUT, KIN, pro, p0, p1, 0.25, None, 0)
        ref = dY[:200000].double().t() @ Xd
        err = float((dW2.double() - ref).norm() / ref.norm())
        fl = 2.0 * rows * NOUT * KIN
        print(f"  {name:40s} {t:8.1f} us  {fl / (t * 1e-6) / 1e12:6.1f} TFLOP/s = {fl / (t * 1e-6) / 157.3e12:.3f} of peak   rel err vs float64 (200k rows) {err:.1e}")
    # layer-0 Toeplitz weight gradient (nshift 8): dW0[256][512]; ConvTranspose1d weight gradient dWct[64][512] (+ bias)
    for S, npos in ((4000, 64), (2048, 125)):
        L = npos - 7
        dU, X = R(S * L, 256), R(S * npos, 64)
        dG, H3 = R(S * npos, 64), R(S * L, 64)
        for form in (0,):
            dW = torch.zeros(256, 512, device="cuda")
            t = timeit(lambda: lib.call("rtfs_wgrad", dU, 256, X, 64, dW, 512, None, S * L, L, npos, 0, 8, 256, 64, 0, None, None, 0.0, None, 0))
            fl = 2.0 * S * L * 256 * 512
            print(f"  Toeplitz dW0, S {S} npos {npos}:        {t:8.1f} us  {fl / (t * 1e-6) / 1e12:6.1f} TFLOP/s (algorithmic) = {fl / (t * 1e-6) / 157.3e12:.3f} of peak")
            dWc, db = torch.zeros(64, 512, device="cuda"), torch.zeros(64, device="cuda")
            t = timeit(lambda: lib.call("rtfs_wgrad", dG, 64, H3, 64, dWc, 512, db, S * npos, npos, L, -7, 8, 64, 64, 0, None, None, 0.0, None, 0))
            fl = 2.0 * S * npos * 64 * 512
            print(f"  Toeplitz dWct, S {S} npos {npos}:       {t:8.1f} us  {fl / (t * 1e-6) / 1e12:6.1f} TFLOP/s (algorithmic) = {fl / (t * 1e-6) / 157.3e12:.3f} of peak")


if __name__ == "__main__":
    main()
