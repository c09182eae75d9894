"""GPU micro-benchmark of the fused SRU layer entry point (rtfs_sru_layer_fwd: input projection on MFMA inside the recurrence) at the two shapes of the
bench workload (freq path S = B x 125 sequences of 57 steps, time path S = B x 64 sequences of 118 steps) + an output checksum for A/B runs.

    python tools/sru_bench.py [dtype: f32|bf16|bf16x3|bf16x6] [B] [form: fp32 only, rtfs_sru_layer_fwd_form's 0-3]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def main(dtype="f32", B=32, form=None):
    terms = {"f32": 0, "bf16": 1, "bf16x3": 3, "bf16x6": 6}[dtype]
    g = torch.Generator().manual_seed(0)
    W = (torch.randn(192, 64, generator=g) * 0.1).cuda()
    wc = (torch.rand(128, generator=g) * 2 - 1).cuda()
    bias = (torch.randn(128, generator=g) * 0.1).cuda()
    for S, L in ((B * 125, 57), (B * 64, 118)):
        h = torch.randn(S, L, 64, generator=g).cuda()
        out = torch.empty_like(h)
        name = "rtfs_sru_layer_fwd" + ("_bf16" if terms else "")
        args = (h, W, wc, bias, 1.0, out, None, None, S, L) + ((terms,) if terms else ())
        if form is not None and not terms:
            name, args = "rtfs_sru_layer_fwd_form", args + (form,)
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
        fl = 2.0 * S * L * 64 * 192
        print(f"{dtype} form {form} S {S} L {L}: median {1e3 * t[len(t) // 2]:.1f} us  min {1e3 * t[0]:.1f} us  {fl / (t[len(t) // 2] * 1e-3) / 1e12:.1f} TFLOP/s (algorithmic)"
              f"   checksum {float(out.double().sum()):.10e} {float(out.double().abs().sum()):.10e}")


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["f32"]), *[int(a) for a in sys.argv[2:4]])
