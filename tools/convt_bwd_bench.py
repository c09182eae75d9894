"""rtfs_convt_bwd_input_form at the bench shape: time per launch, form 0 (library's choice: the fast-FIR kernel) vs 1 (direct).  python tools/convt_bwd_bench.py [B] [T2]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib
B, T2 = (int(a) for a in (sys.argv[1:3] + ["32", "125"][len(sys.argv) - 1:]))
g = torch.Generator().manual_seed(3)
dG = torch.randn(B, T2, 64, 64, generator=g).cuda()
W = (torch.randn(64, 512, generator=g) * 0.05).cuda()
for dim in (4, 3):
    S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
    L = npos - 7
    out = torch.empty(S * L * 64, device="cuda")
    for form in (0, 1):
        for _ in range(3):
            lib.call("rtfs_convt_bwd_input_form", dG, W, out, B, T2, dim, form)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); lib.call("rtfs_convt_bwd_input_form", dG, W, out, B, T2, dim, form); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"convt_bwd_input form {form} dim {dim}: median {1e3 * t[15]:.1f} us  {2.0 * S * L * 512 * 64 / (t[15] * 1e-3) / 1e12:.1f} TFLOP/s (algorithmic)  checksum {float(out.double().abs().sum()):.8e}")
