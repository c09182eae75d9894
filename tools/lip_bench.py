#!/usr/bin/env python
"""Time the HIP lip encoder and print a per-entry-point breakdown.  usage: tools/lip_bench.py [B] [T]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth  # noqa: E402  (weights only; the checker is not timed)
from oracle.lip_ref import lip_inputs  # noqa: E402
from rtfs_net_amd import lib  # noqa: E402
from rtfs_net_amd.models import videomodels  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
m = videomodels.FRCNNVideoModel(print_macs=False)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), salt=3))
m = m.cuda()
m.eval()
x = lip_inputs(B, T).cuda()
with torch.no_grad():
    for _ in range(3):
        y = m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = m(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    m.get_MACs()
    flops = 2e6 * m.macs * B * T / 50
    print(f"lip encoder B={B} T={T}: {ms:.2f} ms/forward = {B * T / ms * 1e3:,.0f} video frames/s, {flops / ms / 1e9:.1f} TFLOP/s fp32")
    lib.profile_begin("*")
    y = m(x)
    torch.cuda.synchronize()
    times, labels = lib.profile_end(), lib.profile_labels()
agg = collections.OrderedDict()
for t, l in zip(times, labels):
    agg.setdefault(l, [0, 0.0])
    agg[l][0] += 1
    agg[l][1] += t
for l, (n, t) in agg.items():
    print(f"  {t * 1e3:9.1f} us  x{n}  {l}")
print(f"  total {sum(times):.2f} ms serialised")
