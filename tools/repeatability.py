#!/usr/bin/env python
"""Run the same RTFS-Net-6, batch-32 forward N times (fp32 and bf16x3) and report the spread between runs (race detector of last resort:
the only run-to-run freedom is the arrival order of the fp64 statistics atomics)."""
import copy
import sys

import torch

sys.path.insert(0, ".")
from rtfs_net_amd import AVNet, synthetic  # noqa: E402

cfg = synthetic.rtfs_audionet(6)
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synthetic.synth_state_dict(model.state_dict()))
model = model.cuda()
mix, _, emb = synthetic.synth_inputs(32, 32000, 50)
mix, emb = mix.cuda(), emb.cuda()
for dtype in ("f32", "bf16x3"):
    model.set_compute_dtype(dtype)
    with torch.no_grad():
        ref = model(mix, emb).double()
        worst = 0.0
        for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
            out = model(mix, emb).double()
            assert torch.isfinite(out).all()
            worst = max(worst, float((out - ref).norm() / ref.norm()))
    print(f"{dtype}: worst run-to-run relative difference {worst:.3e}")
