"""GPU diagnostic: every *_bf16 entry point of the training step against its fp32 sibling on random operands (expected ~1e-5 with terms = 3)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402
from rtfs_net_amd.models.hip_path import pack_bf16  # noqa: E402

g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).cuda()  # noqa: E731
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3

# row GEMMs
for K, N in ((64, 192), (256, 32), (192, 64), (256, 64), (64, 256), (32, 256), (256, 256), (64, 64), (64, 96), (96, 64)):
    M = 5000
    X, W = R(M, K), R(N, K) * 0.1
    Y0, Y1 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    lib.call("rtfs_gemm_rows", X, W, None, Y0, M, K, N, 0)
    lib.call("rtfs_gemm_rows_bf16", X, pack_bf16(W) if T != 6 else W, None, Y1, M, K, N, 0, T)
    print(f"gemm_rows K={K} N={N}: {rel(Y1, Y0):.2e}")
# weight gradients: plain shapes
for NOUT, KIN, pro in ((256, 64, 0), (64, 256, 1), (256, 256, 2), (64, 64, 0), (96, 64, 0), (32, 256, 0), (256, 32, 0), (192, 64, 0)):
    M = 40000
    dY, X = R(M, NOUT), R(M, KIN)
    p0, p1 = R(KIN) * 0.2 + 1, R(KIN) * 0.1
    d0, d1 = torch.zeros(NOUT, KIN, device="cuda"), torch.zeros(NOUT, KIN, device="cuda")
    b0, b1 = torch.zeros(NOUT, device="cuda"), torch.zeros(NOUT, device="cuda")
    args = (dY, NOUT, X, KIN)
    lib.call("rtfs_wgrad", *args, d0, KIN, b0, M, 0, 0, 0, 1, NOUT, KIN, pro, p0, p1, 0.25, None, 0)
    lib.call("rtfs_wgrad_bf16", *args, d1, KIN, b1, M, 0, 0, 0, 1, NOUT, KIN, pro, p0, p1, 0.25, None, 0, T)
    print(f"wgrad NOUT={NOUT} KIN={KIN} pro={pro}: dW {rel(d1, d0):.2e}  dbias {rel(b1, b0):.2e}")
# Toeplitz weight gradient (layer-0 shape: S sequences of L windows over npos positions)
for S, npos in ((200, 64), (70, 125)):
    L = npos - 7
    dU, X = R(S * L, 256), R(S * npos, 64)
    d0, d1 = torch.zeros(256, 512, device="cuda"), torch.zeros(256, 512, device="cuda")
    lib.call("rtfs_wgrad", dU, 256, X, 64, d0, 512, None, S * L, L, npos, 0, 8, 256, 64, 0, None, None, 0.0, None, 0)
    lib.call("rtfs_wgrad_bf16", dU, 256, X, 64, d1, 512, None, S * L, L, npos, 0, 8, 256, 64, 0, None, None, 0.0, None, 0, T)
    print(f"toeplitz wgrad S={S} npos={npos}: {rel(d1, d0):.2e}")
    dG, H3 = R(S * npos, 64), R(S * L, 64)
    d0, d1 = torch.zeros(64, 512, device="cuda"), torch.zeros(64, 512, device="cuda")
    b0, b1 = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    lib.call("rtfs_wgrad", dG, 64, H3, 64, d0, 512, b0, S * npos, npos, L, -7, 8, 64, 64, 0, None, None, 0.0, None, 0)
    lib.call("rtfs_wgrad_bf16", dG, 64, H3, 64, d1, 512, b1, S * npos, npos, L, -7, 8, 64, 64, 0, None, None, 0.0, None, 0, T)
    print(f"convT wgrad S={S} npos={npos}: {rel(d1, d0):.2e}  dbias {rel(b1, b0):.2e}")
# fold / convT input gradients, projection-gateway adjoint
for B, T2, dim in ((2, 125, 4), (2, 125, 3), (3, 40, 3)):
    S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
    L = npos - 7
    dU0, Wf = R(S * L, 256), R(64, 2048) * 0.05
    o0, o1 = torch.zeros(B * T2 * 64 * 64, device="cuda"), torch.zeros(B * T2 * 64 * 64, device="cuda")
    lib.call("rtfs_fold_gemm_bwd", dU0, Wf, o0, B, T2, dim)
    lib.call("rtfs_fold_gemm_bwd_bf16", dU0, pack_bf16(Wf) if T != 6 else Wf, o1, B, T2, dim, T)
    print(f"fold B={B} T2={T2} dim={dim}: {rel(o1, o0):.2e}")
    dG, Wc = R(B * T2 * 64 * 64), R(64, 512) * 0.05
    h0, h1 = torch.zeros(S * L * 64, device="cuda"), torch.zeros(S * L * 64, device="cuda")
    lib.call("rtfs_convt_bwd_input", dG, Wc, h0, B, T2, dim)
    lib.call("rtfs_convt_bwd_input_bf16", dG, pack_bf16(Wc) if T != 6 else Wc, h1, B, T2, dim, T)
    print(f"convt_bwd_input B={B} T2={T2} dim={dim}: {rel(h1, h0):.2e}")
rows = 30000
dy0, WpT, dx, s_in = R(rows, 64), R(256, 64) * 0.1, R(rows, 256), R(rows, 256)
gw, gb = R(256) * 0.2 + 1, R(256) * 0.1
outs = []
for name, Wt, extra in (("rtfs_proj_gateway_bwd", WpT, ()), ("rtfs_proj_gateway_bwd_bf16", pack_bf16(WpT) if T != 6 else WpT, (T,))):
    ds, acc = torch.zeros(rows, 256, device="cuda"), torch.zeros(rows, 256, device="cuda")
    dgw, dgb, dsl = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda"), torch.zeros(1, device="cuda")
    lib.call(name, dy0, Wt, dx, s_in, gw, gb, 0.25, ds, 0, acc, 1, dgw, dgb, dsl, rows, *extra)
    outs.append((ds, acc, dgw, dgb, dsl))
print("proj_gateway_bwd:", [f"{rel(a, b):.2e}" for a, b in zip(outs[1], outs[0])])
torch.cuda.synchronize()
