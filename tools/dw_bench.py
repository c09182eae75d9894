"""GPU micro-benchmark of the depth-wise convolution family at the headline shapes (B = 32, 2 s): the entry points alone, median of 30 launches.
RTFS_HIP_LIB selects the library build (tools/build_variant.sh) for same-box A/B runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return 1e3 * t[len(t) // 2]


def main(B=32, T=251):
    F, F2, T2 = 129, 64, (T - 2) // 2 + 1
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    full = lambda: torch.randn(B * T * F * 64, device=dev, generator=g)  # noqa: E731
    low = lambda: torch.randn(B * T2 * F2 * 64, device=dev, generator=g)  # noqa: E731
    st = lambda n: torch.stack([torch.tensor([0.0, float(n)] + [0.0] * 14, dtype=torch.float64)] * B).to(dev)  # noqa: E731  (mean 0, var 1)
    w = [torch.randn(16 * 64, device=dev, generator=g) * 0.1 for _ in range(4)]
    gam, bet, bias = torch.ones(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    y0, D0, l0, cl = full(), full(), full(), full()
    D1, pooled, g0, gg0 = low(), low(), low(), low()
    outs4 = [low() for _ in range(4)]
    sf, sl = st(T * F * 64), st(T2 * F2 * 64)
    so = [torch.zeros(B, 16, dtype=torch.float64, device=dev) for _ in range(4)]
    res = {}
    res["dwconv_s1<1,2> (y0 -> D0)"] = timeit(lambda: lib.call("rtfs_dwconv_fwd", y0, sf, gam, bet, 0.25, 2, 1, 1, [w[0]], [bias], [D0], [so[0]], B, T, F))
    res["dwconv_trio"] = timeit(lambda: lib.call("rtfs_dwconv_trio_fwd", D0, sf, gam, bet, w[0], l0, so[0], w[1], bias, D1, so[1], pooled, B, T, T2))
    res["dwconv_s1<1,3> (mix -> cl)"] = timeit(lambda: lib.call("rtfs_dwconv_mix_fwd", l0, sf, gam, bet, gg0, sl, gam, bet, g0, sl, gam, bet, 1, [w[0]], [None], [cl], [so[0]],
                                                                 B, T, F, T2, F2))
    res["dwconv_s1<2,3> (low mix)"] = timeit(lambda: lib.call("rtfs_dwconv_mix_fwd", D1, sl, gam, bet, gg0, sl, gam, bet, g0, sl, gam, bet, 2, [w[0], w[1]], [None, None],
                                                               outs4[:2], so[:2], B, T2, F2, T2, F2))
    res["dwconv_s1<4,0> (low x4)"] = timeit(lambda: lib.call("rtfs_dwconv_fwd", g0, None, None, None, 0.0, 0, 1, 4, w, [None] * 4, outs4, so, B, T2, F2))
    res["dwconv_s1<1,1> (low)"] = timeit(lambda: lib.call("rtfs_dwconv_fwd", D1, sl, gam, bet, 0.0, 1, 1, 1, [w[0]], [None], [outs4[0]], [so[0]], B, T2, F2))
    print(lib.library_path())
    for k, v in res.items():
        print(f"  {k:32s} {v:8.1f} us")
    print(f"  {'sum':32s} {sum(res.values()):8.1f} us")


if __name__ == "__main__":
    main()
