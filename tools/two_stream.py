#!/usr/bin/env python
"""Experiment: the batch-32 forward as k sub-batches on k HIP streams (utterances are independent in eval mode), so that one
sub-batch's MFMA-bound kernels share the chip with another's HBM-bound kernels and kernel tails overlap with the next kernel's
head.  Prints ms per step for k = 1, 2, 4 and the difference of the outputs against the one-stream forward.
usage: python tools/two_stream.py [dtype] [steps]"""
import copy
import sys
import time

import torch

sys.path.insert(0, ".")
from rtfs_net_amd import AVNet, synthetic  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 32
cfg = synthetic.rtfs_audionet(6)
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synthetic.synth_state_dict(model.state_dict()))
model = model.cuda()
if dtype != "f32":
    model.set_compute_dtype(dtype)
mix, _, emb = synthetic.synth_inputs(B, 32000, 50)
mix, emb = mix.cuda(), emb.cuda()
# one model object per stream (shared parameters; each owns its prepared-weight cache handle and side stream)
replicas = [model] + [copy.copy(model) for _ in range(3)]


def run(k, stagger=False):
    cur = torch.cuda.current_stream()
    if k == 1:
        return model(mix, emb)
    outs = []
    n = B // k
    for i in range(k):
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(model(mix[i * n:(i + 1) * n], emb[i * n:(i + 1) * n]))
    for s in streams[:k]:
        cur.wait_stream(s)
    return torch.cat(outs, 0)


streams = [torch.cuda.Stream() for _ in range(4)]
with torch.no_grad():
    ref = run(1).double()
    for k in (1, 2, 4, 1, 2, 4):
        for _ in range(3):
            out = run(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run(k)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        err = float((out.double() - ref).norm() / ref.norm())
        print(f"{dtype}: {k} stream(s) x batch {B // k}: {ms:.3f} ms per step, {B * 251 / ms:.1f} k frames/s, rel diff vs one stream {err:.2e}", flush=True)
