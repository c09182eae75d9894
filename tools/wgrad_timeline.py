"""Segment timeline of wgrad_kernel from a -DWG_TIMING build (s_memtime sums per wave: operand LDS reads incl. their wait, the MFMA block's issue,
the next stage's transform + LDS store incl. the wait for its global loads, the load issue two stages ahead, the barrier)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
M = 32 * 251 * 129
for name, NOUT, KIN, rows in (("resid conv dW (64 -> 256)", 256, 64, M), ("SRU layer dW (64 -> 192)", 192, 64, 2048 * 118)):
    dY, X = R(rows, NOUT), R(rows, KIN)
    dW = torch.zeros(NOUT, KIN, device="cuda")
    dump = torch.zeros(8 * 1024 * 1024, device="cuda")  # plays dbias: must be large enough for grid x 4 waves x 6 int64
    for _ in range(2):
        lib.call("rtfs_wgrad", dY, NOUT, X, KIN, dW, KIN, dump, rows, 0, 0, 0, 1, NOUT, KIN, 0, None, None, 0.25, None, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.call("rtfs_wgrad", dY, NOUT, X, KIN, dW, KIN, dump, rows, 0, 0, 0, 1, NOUT, KIN, 0, None, None, 0.25, None, 0)
    e1.record()
    torch.cuda.synchronize()
    wall_us = 1e3 * e0.elapsed_time(e1)
    raw = dump.view(torch.int64)[: 4096 * 4 * 8].view(-1, 4, 8).cpu()
    spans = []
    for x in range(8):  # workgroup b runs on XCD b % 8 (observed): s_memtime bases differ per XCD
        r = raw[x::8].reshape(-1, 8)
        r = r[r[:, 5] > 0]
        spans.append(int((r[:, 7] + r[:, 6]).max() - r[:, 7].min()))
    print(f"  wall {wall_us:.1f} us; per-XCD span of s_memtime stamps {min(spans)} .. {max(spans)} ticks -> {max(spans) / wall_us:.0f} ticks per us")
    v = dump.view(torch.int64)[: 4096 * 4 * 8].view(-1, 8).cpu().double()
    v = v[v[:, 5] > 0]
    per = v[:, :5] / v[:, 5:6]
    names = ("operand reads", "MFMA issue", "xform + LDS store (+ load wait)", "load issue", "barrier")
    print(f"{name}: {len(v)} waves, stages per workgroup {v[:,5].median():.0f}; cycles per stage: " + ", ".join(f"{n} {per[:, j].median():.0f}" for j, n in enumerate(names)) + f"; total {per.sum(1).median():.0f}; whole workgroup {v[:, 6].median():.0f} cycles (loop {v[:, :5].sum(1).median():.0f}); kernel span {v[:, 7].max() - v[:, 7].min() + v[:, 6].median():.0f} ticks (one XCD clock domain only if stamps agree)")
