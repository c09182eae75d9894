"""Numerics prototype (CPU, torch fp32 vs float64) of the 2-parallel fast-FIR form of the SRU layer-0 GEMM proposed in DESIGN.md section 7.

Direct form (rnn_layers.py:59-105 as the kernels compute it): U[l] = sum_{k<8} W_k x[l+k], x [npos][64], W_k [256][64], l = 0 .. L-1, L = npos - 7.
Fast form: e[m] = x[2m], u[m] = x[2m-1] (u[0] = 0);  P = W_e * e, Q = W_o * u, S = (W_e + W_o) * (e + u)  (4-tap correlations, W_e[j] = W_2j, W_o[j] = W_2j+1)
           U[2m] = P[m] + Q[m+1],  U[2m-1] = S[m] - P[m] - Q[m]          -> 3 half-length products instead of 4: 0.75x the multiply-adds.
Prints the relative L2 / max errors of both fp32 forms against float64 for LayerNorm-ed inputs and the synthetic weights' scale."""
import torch

torch.manual_seed(0)


def direct(x, W):  # x [S][npos][64], W [8][256][64]
    L = x.shape[1] - 7
    return sum(x[:, k:k + L] @ W[k].t() for k in range(8))


def corr4(x, W4, n):  # 4-tap correlation: out[m] = sum_j W4[j] x[m+j], m = 0 .. n-1 (x zero-padded at the end)
    pad = torch.zeros(x.shape[0], max(0, n + 3 - x.shape[1]), x.shape[2], dtype=x.dtype)
    xp = torch.cat([x, pad], 1)
    return sum(xp[:, j:j + n] @ W4[j].t() for j in range(4))


def fast(x, W):
    S_, npos, _ = x.shape
    L = npos - 7
    e = x[:, 0::2]                                                       # e[m] = x[2m]
    u = torch.cat([torch.zeros_like(x[:, :1]), x[:, 1::2]], 1)           # u[m] = x[2m-1], u[0] = 0
    n = max(e.shape[1], u.shape[1])
    e = torch.cat([e, torch.zeros(S_, n - e.shape[1], 64, dtype=x.dtype)], 1)
    u = torch.cat([u, torch.zeros(S_, n - u.shape[1], 64, dtype=x.dtype)], 1)
    We, Wo = W[0::2], W[1::2]
    M = (L + 1) // 2 + 1
    P, Q, S = corr4(e, We, M + 1), corr4(u, Wo, M + 1), corr4(e + u, We + Wo, M + 1)
    U = torch.empty(S_, L, 256, dtype=x.dtype)
    me = torch.arange(0, (L + 1) // 2)            # even outputs 2m <= L-1
    U[:, 0::2] = P[:, me] + Q[:, me + 1]
    mo = torch.arange(1, L // 2 + 1)              # odd outputs 2m-1 <= L-1
    U[:, 1::2] = S[:, mo] - P[:, mo] - Q[:, mo]
    return U


for npos in (64, 125, 33):
    x64 = torch.randn(48, npos, 64, dtype=torch.float64)
    x64 = (x64 - x64.mean(-1, keepdim=True)) / x64.std(-1, keepdim=True)          # LayerNorm-ed rows, as the kernel's input
    W64 = torch.randn(8, 256, 64, dtype=torch.float64) * (512 ** -0.5)
    ref = direct(x64, W64)
    assert torch.allclose(fast(x64, W64), ref, rtol=1e-12, atol=1e-12), "identity wrong"
    d32, f32 = direct(x64.float(), W64.float()).double(), fast(x64.float(), W64.float()).double()
    rel = lambda a: float((a - ref).norm() / ref.norm())  # noqa: E731
    print(f"npos {npos:4d}: direct fp32 rel L2 {rel(d32):.2e} max {float((d32 - ref).abs().max()):.2e} | fast-FIR fp32 rel L2 {rel(f32):.2e} max {float((f32 - ref).abs().max()):.2e}"
          f" | odd outputs only {float((f32[:, 1::2] - ref[:, 1::2]).norm() / ref[:, 1::2].norm()):.2e}")
