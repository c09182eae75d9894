"""rtfs_fold_gemm_bwd (input gradient of the layer-0 GEMM) against float64, both dual paths; time per launch and relative L2 error.
    python tools/fold_bwd_bench.py [B] [T2]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def main(B=32, T2=125, reps=30):
    g = torch.Generator().manual_seed(0)
    W0 = (torch.randn(256, 512, generator=g) * 0.05).cuda()
    Wf = W0.view(256, 8, 64).flip(1).permute(2, 1, 0).reshape(64, 2048).contiguous()
    for dim in (4, 3):
        S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
        L = npos - 7
        dU = torch.randn(S, L, 256, generator=g).cuda()
        want = torch.zeros(S, npos, 64, dtype=torch.float64, device="cuda")
        for k in range(8):
            want[:, k:k + L] += dU.double() @ W0.double()[:, 64 * k:64 * k + 64]
        want = want.view(B, T2, 64, 64) if dim == 4 else want.view(B, 64, T2, 64).permute(0, 2, 1, 3)
        for variant in (0,):
            dxn = torch.full((B, T2, 64, 64), float("nan"), device="cuda")
            for _ in range(3):
                lib.call("rtfs_fold_gemm_bwd", dU, Wf, dxn, B, T2, dim)
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in ev:
                a.record()
                lib.call("rtfs_fold_gemm_bwd", dU, Wf, dxn, B, T2, dim)
                b.record()
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in ev)
            err = float((dxn.double() - want).norm() / want.norm())
            bad = int((~torch.isfinite(dxn)).sum())
            rows = ((dxn.double() - want).reshape(-1, 64).norm(dim=-1) / want.reshape(-1, 64).norm(dim=-1)) if bad == 0 else None
            fl = 2.0 * S * L * 512 * 256
            print(f"B {B} T2 {T2} dim {dim}: median {1e3 * t[len(t) // 2]:.1f} us  min {1e3 * t[0]:.1f} us  {fl / (t[len(t) // 2] * 1e-3) / 1e12:.1f} TFLOP/s (algorithmic)  "
                  f"rel L2 vs float64 {err:.2e}  worst row {float(rows.max()) if rows is not None else float('nan'):.2e}  non-finite {bad}", flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
