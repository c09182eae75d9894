"""GPU: determinism of the dual-path launch sequence unfold GEMM -> layer-0 scan -> fused SRU layer in plain bf16, launched back to back
(no synchronisation in between) - the combination tools/bf16_bisect.py points at.  usage: python tools/seq_det.py [terms_unfold] [terms_layer] [sync 0|1] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib  # noqa: E402
from rtfs_net_amd.models.hip_path import pack_bf16  # noqa: E402

tu, tl, sync, N = [int(a) for a in (sys.argv[1:5] + ["1", "1", "0", "60"][len(sys.argv) - 1:])]
g = torch.Generator().manual_seed(0)
B, T2, dim = 32, 125, 4
S, npos = B * T2, 64
L = npos - 7
G = torch.randn(B, T2, 64, 64, generator=g).cuda()
g64, b64 = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
W0 = pack_bf16((torch.randn(256, 512, generator=g) * 0.05).cuda())
wc, bs = (torch.rand(128, generator=g) * 2 - 1).cuda(), (torch.randn(128, generator=g) * 0.3).cuda()
W1 = (torch.randn(192, 64, generator=g) * 0.3).cuda()
U = torch.empty(S * L * 256, device="cuda")
h0 = torch.empty(S * L * 64, device="cuda")
h1 = torch.empty(S * L * 64, device="cuda")
refs, bad = None, [0, 0, 0]
for it in range(N):
    U.fill_(float("nan")), h0.fill_(float("nan")), h1.fill_(float("nan"))
    torch.cuda.synchronize()
    lib.call("rtfs_dp_unfold_gemm_fwd_bf16", G, g64, b64, W0, U, B, T2, dim, 0, tu)
    if sync:
        torch.cuda.synchronize()
    lib.call("rtfs_sru_scan_fwd", U, None, wc, bs, 1.0, h0, S, L, 4)
    if sync:
        torch.cuda.synchronize()
    lib.call("rtfs_sru_layer_fwd_bf16", h0, W1, wc, bs, 0.9, h1, None, None, S, L, tl)
    torch.cuda.synchronize()
    cur = (U.clone(), h0.clone(), h1.clone())
    if refs is None:
        refs = cur
    else:
        for j in range(3):
            bad[j] += int(not torch.equal(cur[j].view(torch.int32), refs[j].view(torch.int32)))
print(f"unfold terms {tu}, layer terms {tl}, sync {sync}: U0 differs {bad[0]}, scan output differs {bad[1]}, layer output differs {bad[2]} of {N - 1}", flush=True)
