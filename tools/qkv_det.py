"""GPU diagnostic: bitwise run-to-run determinism of rtfs_attn_qkv_fwd(_bf16) in isolation.

usage: python tools/qkv_det.py [runs] [terms ...]   (QD_SYNC=1: device synchronise around every launch)
Prints, per precision mode, how many runs differ from the first one in Q, K, V and the pre-activation output, and for the first
differing runs the differing elements as (start, stride, count) ranges with the number of exact zeros / NaNs in the output.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import lib  # noqa: E402
from rtfs_net_amd.models.hip_path import pack_bf16  # noqa: E402


def ranges(idx):
    out, i = [], 0
    while i < len(idx) and len(out) < 12:
        j, step = i + 1, None
        while j < len(idx) and (step is None or idx[j] - idx[j - 1] == step):
            step = idx[j] - idx[j - 1]
            j += 1
        out.append((idx[i], step, j - i))
        i = j
    return out


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    modes = [int(x) for x in sys.argv[2:]] or [0, 3, 1, 6]
    sync = os.environ.get("QD_SYNC") == "1"
    B, T2 = 32, 125
    g = torch.Generator(device="cuda").manual_seed(0)
    G = torch.randn(B * T2 * 64 * 64, device="cuda", generator=g)
    W = torch.randn(96, 64, device="cuda", generator=g) * 0.1
    bias = torch.randn(96, device="cuda", generator=g) * 0.1
    slope = torch.full((96,), 0.25, device="cuda")
    gq, bq = torch.ones(4, 256, device="cuda"), torch.zeros(4, 256, device="cuda")
    gv, bv = torch.ones(4, 1024, device="cuda"), torch.zeros(4, 1024, device="cuda")
    bad = 0
    for terms in modes:
        Wk = pack_bf16(W) if terms in (1, 3) else W
        name = "rtfs_attn_qkv_fwd" + ("_bf16" if terms else "")
        ref, nd = None, [0, 0, 0, 0]
        for it in range(runs):
            Q = torch.full((B * 4 * T2 * 256,), float("nan"), device="cuda")
            K = torch.full_like(Q, float("nan"))
            V = torch.full((B * 4 * T2 * 1024,), float("nan"), device="cuda")
            Y = torch.full((B * T2 * 64 * 96,), float("nan"), device="cuda")
            if sync:
                torch.cuda.synchronize()
            lib.call(name, G, Wk, bias, slope, gq, bq, gq, bq, gv, bv, Q, K, V, Y, B, T2, *((terms,) if terms else ()))
            if sync:
                torch.cuda.synchronize()
            cur = (Q, K, V, Y)
            if ref is None:
                ref = cur
                print("terms", terms, "first run: zeros", [int((x == 0).sum()) for x in cur], "nans", [int(x.isnan().sum()) for x in cur])
                continue
            for j in range(4):
                ne = cur[j].view(torch.int32) != ref[j].view(torch.int32)
                if bool(ne.any()):
                    nd[j] += 1
                    if nd[j] <= 3:
                        d = ne.nonzero().flatten().tolist()
                        per = 96 if j == 3 else (1024 if j == 2 else 256)
                        odd = sum(((x // per) % T2 + ((x // per) // T2 // (1 if j == 3 else 4)) * T2) % 2 for x in d) if j < 3 else 0
                        print("terms", terms, "run", it, "QKVY"[j], "diffs", len(d), "in odd tokens", odd, "ranges", ranges(d)[:6], "zeros cur/ref",
                              int((cur[j] == 0).sum()), int((ref[j] == 0).sum()))
        print("terms", terms, "runs differing (Q, K, V, Ypre):", nd)
        bad += sum(nd)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
