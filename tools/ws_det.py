"""GPU: run-to-run bit identity of the weight-stationary entry points in every precision (same inputs, N back-to-back launches each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib  # noqa: E402
from rtfs_net_amd.models.hip_path import pack_bf16  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(0)
B, T, T2 = 32, 251, 125
TF = T * 129
x = torch.randn(B, TF, 256, generator=g).cuda()
emb = torch.randn(B, TF, 256, generator=g).cuda()
W = (torch.randn(256, 256, generator=g) * 0.06).cuda()
bias = (torch.randn(256, generator=g) * 0.1).cuda()
gamma, beta = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
stats = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
stats[:, 0], stats[:, 1] = x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))
G = torch.randn(B, T2, 64, 64, generator=g).cuda()
g64, b64 = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
W0 = (torch.randn(256, 512, generator=g) * 0.05).cuda()
Wc = (torch.randn(64, 512, generator=g) * 0.05).cuda()
bc = (torch.randn(64, generator=g) * 0.1).cuda()


def check(name, fn, out):
    ref = None
    bad = 0
    worst = 0.0
    for _ in range(N):
        out.fill_(float("nan"))
        fn(out)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out.view(torch.int32), ref.view(torch.int32)):
            bad += 1
            worst = max(worst, float((out.double() - ref.double()).norm() / ref.double().norm()))
    print(f"{name}: {bad} of {N - 1} launches differ from the first (max rel {worst:.2e})", flush=True)


for terms in (0, 1, 3):
    sfx, ta = ("_bf16", (terms,)) if terms else ("", ())
    Wk, W0k, Wck = (pack_bf16(W), pack_bf16(W0), pack_bf16(Wc)) if terms else (W, W0, Wc)
    y = torch.empty(B * TF * 256, device="cuda")
    check(f"terms {terms} bottleneck", lambda o: lib.call("rtfs_bottleneck_fwd" + sfx, x, stats, gamma, beta, Wk, bias, o, B, TF, *ta), y)
    check(f"terms {terms} mask", lambda o: lib.call("rtfs_mask_fwd" + sfx, x, 0.25, Wk, bias, emb, o, None, B, TF, *ta), y)
    check(f"terms {terms} rows", lambda o: lib.call("rtfs_gemm_rows" + sfx, x.view(-1, 256), Wk, None, o, B * TF, 256, 256, 0, *ta), y)
    for dim in (4, 3):
        S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
        L = npos - 7
        U = torch.empty(S * L * 256, device="cuda")
        check(f"terms {terms} unfold dim {dim}", lambda o: lib.call("rtfs_dp_unfold_gemm_fwd" + sfx, G, g64, b64, W0k, o, B, T2, dim, 0, *ta), U)
        H3 = torch.randn(S, L, 64, generator=torch.Generator().manual_seed(dim)).cuda()
        G0 = G.clone()
        Gw = torch.empty_like(G)

        def convt(o):
            o.copy_(G0)
            lib.call("rtfs_dp_convt_fwd" + sfx, H3, Wck, bc, o, B, T2, dim, *ta)

        check(f"terms {terms} convt dim {dim}", convt, Gw)
