"""Layer-0 unfold GEMM in the six-term split (bf16x6): the weight-stationary kernel (variant 0 at large batch) against the LDS-staged kernel
(variant 2), the fp32 fast-FIR kernel and float64 (torch on the GPU), both dual paths; time per launch and relative L2 error of U0.
    python tools/ws6_check.py [B] [T2]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def main(B=32, T2=125, reps=30):
    g = torch.Generator().manual_seed(0)
    G = torch.randn(B, T2, 64, 64, generator=g).cuda()
    gamma, beta = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    W = (torch.randn(256, 512, generator=g) * 0.05).cuda()
    x = G.double()
    xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    for dim in (4, 3):
        seqs = xn.reshape(B * T2, 64, 64) if dim == 4 else xn.permute(0, 2, 1, 3).reshape(B * 64, T2, 64)
        S, npos = seqs.shape[0], seqs.shape[1]
        L = npos - 7
        want = torch.zeros(S, L, 256, dtype=torch.float64, device="cuda")
        for k in range(8):
            want += seqs[:, k:k + L] @ W.double()[:, 64 * k:64 * k + 64].t()
        for name, terms, variant in (("bf16x6 ws", 6, 0), ("bf16x6 lds", 6, 2), ("f32 ffa", 0, 0)):
            U = torch.full((S * L * 256,), float("nan"), device="cuda")

            def launch():
                if terms:
                    lib.call("rtfs_dp_unfold_gemm_fwd_bf16", G, gamma, beta, W, U, B, T2, dim, variant, terms)
                else:
                    lib.call("rtfs_dp_unfold_gemm_fwd", G, gamma, beta, W, U, B, T2, dim, variant)
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in ev:
                a.record()
                launch()
                b.record()
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in ev)
            Uv = U.view(S, L, 256).double()
            err = float((Uv - want).norm() / want.norm())
            bad = int((~torch.isfinite(Uv)).sum())
            worst_row = float(((Uv - want).norm(dim=-1) / want.norm(dim=-1)).max()) if bad == 0 else float("nan")
            fl = 2.0 * S * L * 512 * 256
            print(f"B {B} T2 {T2} dim {dim} {name}: median {1e3 * t[len(t) // 2]:.1f} us  min {1e3 * t[0]:.1f} us  {fl / (t[len(t) // 2] * 1e-3) / 1e12:.1f} TFLOP/s (algorithmic)  "
                  f"rel L2 vs float64 {err:.2e}  worst row {worst_row:.2e}  non-finite {bad}", flush=True)
            if bad:
                nz = torch.nonzero(~torch.isfinite(Uv).all(-1))
                print("   first non-finite rows (s, l):", nz[:8].tolist())
            elif err > 1e-5:
                rowerr = ((Uv - want).norm(dim=-1) / want.norm(dim=-1))
                nz = torch.nonzero(rowerr > 1e-4)
                print("   bad rows:", nz.shape[0], "first (s, l):", nz[:12].tolist())
                s0, l0 = nz[0].tolist()
                colerr = (Uv[s0, l0] - want[s0, l0]).abs()
                print("   bad cols of first bad row:", torch.nonzero(colerr > 1e-4 * want[s0, l0].abs().max()).flatten()[:40].tolist())


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
