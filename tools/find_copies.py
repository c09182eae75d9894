#!/usr/bin/env python
"""Where do the small device copies of one inference forward come from?  (torch.profiler, python stacks of every Memcpy / copy_ op)"""
import copy
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
from rtfs_net_amd import AVNet, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = synthetic.rtfs_audionet(6)
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synthetic.synth_state_dict(model.state_dict()))
model = model.cuda()
mix, _, emb = synthetic.synth_inputs(B, 32000, 50)
mix, emb = mix.cuda(), emb.cuda()
with torch.no_grad():
    for _ in range(2):
        model(mix, emb)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(mix, emb)
        torch.cuda.synchronize()
rows = {}
for e in prof.events():
    n = e.name
    if "emcpy" in n or n in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_"):
        st = [s for s in (e.stack or []) if "rtfs_net_amd" in s]
        key = (n, st[0] if st else "?")
        rows[key] = rows.get(key, 0) + 1
for (n, s), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:28s} {s}")
