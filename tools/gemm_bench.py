"""GPU micro-benchmark of the layer-0 unfold GEMM entry point (both dual paths at the bench shape) + an output checksum for A/B runs.

    python tools/gemm_bench.py [dtype: f32|bf16|bf16x3] [B] [T2] [variant: 0 launcher's choice | 1 per-sequence tiles | 2 LDS-staged flattened tiles]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402
from rtfs_net_amd.models.hip_path import COMPUTE_DTYPES, pack_bf16  # noqa: E402


def main(dtype="f32", B=32, T2=125, variant=0):
    prec = COMPUTE_DTYPES[dtype]
    g = torch.Generator().manual_seed(0)
    G = torch.randn(B, T2, 64, 64, generator=g).cuda()
    gamma, beta = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    W = (torch.randn(256, 512, generator=g) * 0.05).cuda()
    Wk = pack_bf16(W) if prec in (1, 3) else W
    for dim in (4, 3):
        S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
        L = npos - 7
        U = torch.empty(S * L * 256, device="cuda")
        name = "rtfs_dp_unfold_gemm_fwd" + ("_bf16" if prec else "")
        args = (G, gamma, beta, Wk, U, B, T2, dim, variant) + ((prec,) if prec else ())
        for _ in range(3):
            lib.call(name, *args)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record()
            lib.call(name, *args)
            b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        fl = 2.0 * S * L * 512 * 256
        print(f"{dtype} variant {variant} dim {dim}: median {1e3 * t[len(t) // 2]:.1f} us  min {1e3 * t[0]:.1f} us  {fl / (t[len(t) // 2] * 1e-3) / 1e12:.1f} TFLOP/s   checksum {float(U.double().sum()):.10e} {float(U.double().abs().sum()):.10e}")


def convt(B=32, T2=125, form=0):
    """rtfs_dp_convt_fwd_form (ConvTranspose1d + bias + residual, in place on G; form 0 = library's choice, 1 = direct 8-tap kernels) at the bench shape:
    time + checksum of one application"""
    g = torch.Generator().manual_seed(1)
    for dim in (4, 3):
        S, npos = (B * T2, 64) if dim == 4 else (B * 64, T2)
        L = npos - 7
        H3 = torch.randn(S, L, 64, generator=g).cuda()
        W = (torch.randn(64, 512, generator=g) * 0.05).cuda()
        bias = (torch.randn(64, generator=g) * 0.1).cuda()
        G0 = torch.randn(B, T2, 64, 64, generator=g).cuda()
        G = G0.clone()
        lib.call("rtfs_dp_convt_fwd_form", H3, W, bias, G, B, T2, dim, form)
        chk = (float(G.double().sum()), float(G.double().abs().sum()))
        # float64 check of a few sequences: y[n] = sum_k' hpad[n + k'] . W'[k'] + bias + G0[n], k index = k' * 64 + channel
        Gs = (G if dim == 4 else G.permute(0, 2, 1, 3)).reshape(S, npos, 64)
        G0s = (G0 if dim == 4 else G0.permute(0, 2, 1, 3)).reshape(S, npos, 64)
        err = 0.0
        for s_ in (0, 1, S // 2, S - 1):
            hp = torch.zeros(npos + 7 + 7, 64, dtype=torch.float64)
            hp[7:7 + L] = H3[s_].double().cpu()
            win = torch.stack([hp[k:k + npos] for k in range(8)], 1).reshape(npos, 512)
            ref = win @ W.double().cpu().t() + bias.double().cpu() + G0s[s_].double().cpu()
            err = max(err, float((Gs[s_].double().cpu() - ref).abs().max() / ref.abs().max()))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record()
            lib.call("rtfs_dp_convt_fwd_form", H3, W, bias, G, B, T2, dim, form)
            b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        fl = 2.0 * S * npos * 512 * 64
        print(f"convt form {form} dim {dim}: median {1e3 * t[len(t) // 2]:.1f} us  min {1e3 * t[0]:.1f} us  {fl / (t[len(t) // 2] * 1e-3) / 1e12:.1f} TFLOP/s   checksum {chk[0]:.10e} {chk[1]:.10e}   max rel err vs float64 (4 sequences) {err:.2e}")


if __name__ == "__main__":
    if sys.argv[1:2] == ["convt"]:
        convt(*[int(a) for a in sys.argv[2:5]])
        sys.exit(0)
    main(*(sys.argv[1:2] or ["f32"]), *[int(a) for a in sys.argv[2:5]])
