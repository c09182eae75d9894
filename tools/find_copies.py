#!/usr/bin/env python
"""What else runs in one inference forward besides the rtfs_* kernels?  torch.profiler: every aten op and every device activity that is not
an rtfs kernel, with counts (the small `__amd_rocclr_copyBuffer` launches of the kernel trace show up here with their origin)."""
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
    if n.startswith("void rtfs::") or n.startswith("rtfs::"):
        continue
    st = [s for s in (e.stack or []) if "rtfs_net_amd" in s]
    key = (str(e.device_type).split(".")[-1], n[:60], st[0][-70:] if st else "")
    rows[key] = rows.get(key, 0) + 1
for (d, n, s), c in sorted(rows.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d}  {d:5s} {n:60s} {s}")
