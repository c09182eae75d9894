"""GPU micro-benchmark of the three attention entry points (MultiHeadSelfAttention2D, layers/attention.py:149-189) at the bench shape + checksums."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return 1e3 * t[len(t) // 2]


def main(B=32, T2=125):
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    Q, K, V = rnd(B, 4, T2, 256), rnd(B, 4, T2, 256), rnd(B, 4, T2, 1024)
    O = torch.empty(B, T2, 64, 64, device="cuda")
    t = timeit(lambda: lib.call("rtfs_attn_core_fwd", Q, K, V, O, None, B, T2))
    print(f"{lib.library_path()}: attn_core B {B} T2 {T2}: {t:.1f} us   checksum {float(O.double().sum()):.10e} {float(O.double().abs().sum()):.10e}")


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
