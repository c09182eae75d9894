import os, sys, torch
sys.path.insert(0, "/root/repo")
from rtfs_net_amd import lib
from rtfs_net_amd.models.hip_path import COMPUTE_DTYPES, pack_bf16
for dtype in (sys.argv[1:] or ("f32", "bf16x3", "bf16")):
    prec = COMPUTE_DTYPES[dtype]
    g = torch.Generator().manual_seed(0)
    B, T2 = 32, 125
    G = torch.randn(B, T2, 64, 64, generator=g).cuda()
    gamma, beta = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    W = (torch.randn(256, 512, generator=g) * 0.05).cuda()
    Wk = pack_bf16(W) if prec in (1, 3) else W
    for zero in (False, True):
        Gi = torch.zeros_like(G) if zero else G
        S, L = B * T2, 57
        U = torch.empty(S * L * 256, device="cuda")
        name = "rtfs_dp_unfold_gemm_fwd" + ("_bf16" if prec else "")
        args = (Gi, gamma, beta, Wk, U, B, T2, 4, 0) + ((prec,) if prec else ())
        for _ in range(5): lib.call(name, *args)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); lib.call(name, *args); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ts.sort()
        v = U.view(torch.int64)[:512].cpu().view(256, 2)
        ticks, tiles = v[:, 0].double(), v[:, 1].double()
        print(f"{dtype} zero-input={zero}: wall median {1e3*ts[10]:.1f} us; per-workgroup s_memtime ticks median {ticks.median():.0f} (max {ticks.max():.0f}) for {tiles.median():.0f} tiles"
              f" -> {ticks.median()/tiles.median():.0f} ticks per tile; ticks / wall = {ticks.max()/(1e3*ts[10]):.0f} per us")
