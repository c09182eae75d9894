"""GPU micro-benchmark of the bandwidth-bound BACKWARD kernels of one RTFS block at the headline shapes (B = 32, 2 s): entry points alone,
median of 30 launches, with the algorithmic bytes each moves.  RTFS_HIP_LIB selects the library build for same-box A/B runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib  # noqa: E402
from tools.dw_bench import timeit  # noqa: E402


def main(B=32, T=251):
    F, F2, T2 = 129, 64, (T - 2) // 2 + 1
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    nf, nl = B * T * F * 64, B * T2 * F2 * 64
    full = lambda: torch.randn(nf, device=dev, generator=g)  # noqa: E731
    low = lambda: torch.randn(nl, device=dev, generator=g)  # noqa: E731
    st = lambda n: torch.stack([torch.tensor([0.0, float(n)] + [0.0] * 14, dtype=torch.float64)] * B).to(dev)  # noqa: E731
    w = torch.randn(16 * 64, device=dev, generator=g) * 0.1
    gam, bet = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    a, b, c, d = full(), full(), full(), full()
    la, lb, lc, ld = low(), low(), low(), low()
    sf, sl = st(T * F * 64), st(T2 * F2 * 64)
    red = torch.zeros(B, 16, dtype=torch.float64, device=dev)
    red3 = torch.zeros(3, B, 16, dtype=torch.float64, device=dev)
    dgb = [torch.zeros(64, device=dev) for _ in range(6)]
    le, lf, lg_ = low(), low(), low()
    dg, db, dsl = torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(1, device=dev)
    dW, dbias = torch.zeros(1024, device=dev), torch.zeros(64, device=dev)
    FB, LB = 4 * nf, 4 * nl
    res = {}

    def run(name, nbytes, fn):
        res[name] = (timeit(fn), nbytes)

    run("gln_bwd_reduce full act0", 2 * FB, lambda: lib.call("rtfs_gln_bwd_reduce", a, b, sf, gam, bet, 0, 0.0, red, dg, db, None, B, T * F, 64))
    run("gln_bwd_reduce full act1", 2 * FB, lambda: lib.call("rtfs_gln_bwd_reduce", a, b, sf, gam, bet, 1, 0.25, red, dg, db, dsl, B, T * F, 64))
    run("gln_bwd_apply  full act0", 3 * FB, lambda: lib.call("rtfs_gln_bwd_apply", a, b, sf, gam, bet, 0, 0.0, red, c, 0, B, T * F, 64))
    run("gln_bwd_apply  full act1", 3 * FB, lambda: lib.call("rtfs_gln_bwd_apply", a, b, sf, gam, bet, 1, 0.25, red, c, 0, B, T * F, 64))
    run("gln_bwd_reduce low", 2 * LB, lambda: lib.call("rtfs_gln_bwd_reduce", la, lb, sl, gam, bet, 0, 0.0, red, dg, db, None, B, T2 * F2, 64))
    run("gln_bwd_apply  low", 3 * LB, lambda: lib.call("rtfs_gln_bwd_apply", la, lb, sl, gam, bet, 0, 0.0, red, lc, 0, B, T2 * F2, 64))
    run("mix_gln_bwd full", 5 * FB + 3 * LB, lambda: lib.call("rtfs_mix_gln_bwd", a, b, sf, gam, bet, la, sl, gam, bet, ld, sl, gam, bet, c, lb, lc, red3, dgb, B, T, F, T2, F2))
    run("mix_bwd full (unfused part)", 4 * FB + 3 * LB, lambda: lib.call("rtfs_mix_bwd", a, b, sf, gam, bet, la, sl, gam, bet, c, lb, lc, B, T, F, T2, F2))
    run("mix_gln_bwd low", 8 * LB, lambda: lib.call("rtfs_mix_gln_bwd", la, lb, sl, gam, bet, lc, sl, gam, bet, ld, sl, gam, bet, le, lf, lg_, red3, dgb, B, T2, F2, T2, F2))
    run("dwconv_bwd_weight full mode0", 2 * FB, lambda: lib.call("rtfs_dwconv_bwd_weight", a, b, None, None, None, 0.0, 0, 1, dW, None, B, T, F))
    run("dwconv_bwd_weight full mode1", 2 * FB, lambda: lib.call("rtfs_dwconv_bwd_weight", a, b, sf, gam, bet, 0.0, 1, 1, dW, None, B, T, F))
    run("dwconv_bwd_weight full mode2", 2 * FB, lambda: lib.call("rtfs_dwconv_bwd_weight", a, b, sf, gam, bet, 0.25, 2, 1, dW, dbias, B, T, F))
    run("dwconv_bwd_weight stride2", FB + LB, lambda: lib.call("rtfs_dwconv_bwd_weight", la, b, sf, gam, bet, 0.0, 1, 2, dW, dbias, B, T, F))
    run("dwconv_bwd_weight low mode0", 2 * LB, lambda: lib.call("rtfs_dwconv_bwd_weight", la, lb, None, None, None, 0.0, 0, 1, dW, None, B, T2, F2))
    run("dwconv_bwd_input full =", 2 * FB, lambda: lib.call("rtfs_dwconv_bwd_input", a, w, c, 0, 1, B, T, F))
    run("dwconv_bwd_input full +=", 3 * FB, lambda: lib.call("rtfs_dwconv_bwd_input", a, w, c, 1, 1, B, T, F))
    run("dwconv_bwd_input stride2 +=", 2 * FB + LB, lambda: lib.call("rtfs_dwconv_bwd_input", la, w, c, 1, 2, B, T, F))
    run("dwconv_bwd_input low =", 2 * LB, lambda: lib.call("rtfs_dwconv_bwd_input", la, w, lc, 0, 1, B, T2, F2))
    run("dwconv_bwd_input low +=", 3 * LB, lambda: lib.call("rtfs_dwconv_bwd_input", la, w, lc, 1, 1, B, T2, F2))
    run("pool_bwd", 2 * FB + LB, lambda: lib.call("rtfs_pool_bwd", la, c, B, T, T2))
    run("expand_fwd", 3 * FB + 2 * LB, lambda: lib.call("rtfs_expand_fwd", a, sf, gam, bet, b, sf, gam, bet, la, sl, gam, bet, lb, sl, gam, bet, c, B, T, T2))
    print(lib.library_path())
    for k, (us, nb) in res.items():
        print(f"  {k:32s} {us:8.1f} us   {nb / 1e6:7.0f} MB   {nb / us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
