#!/usr/bin/env python
"""Per-call-site time of one step (HIP events around EVERY C-ABI launch; serialising, so only the split is meaningful).
usage: python tools/train_breakdown.py [train|infer] [batch] [layers]"""
import collections
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import synth  # noqa: E402  (synthetic weights/inputs only)
from rtfs_net_amd import AVNet, lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
R = int(sys.argv[3]) if len(sys.argv) > 3 else 6
L, Tv = 32000, 50
dev = torch.device("cuda:0")
model = AVNet(print_macs=False, **copy.deepcopy(synth.rtfs_audionet(R)))
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.to(dev).train(mode == "train")
mix, _, emb = synth.synth_inputs(B, L, Tv)
mix, emb = mix.to(dev), emb.to(dev)


def step():
    if mode == "train":
        model.zero_grad(set_to_none=True)
        model(mix, emb).square().mean().backward()
    else:
        with torch.no_grad():
            model(mix, emb)


step()
torch.cuda.synchronize()
lib.profile_begin("*")
step()
ms = lib.profile_end()
agg = collections.defaultdict(lambda: [0, 0.0])
for lab, t in zip(lib.profile_labels(), ms):
    agg[lab][0] += 1
    agg[lab][1] += t
tot = sum(ms)
print(f"# {mode} B={B} R={R}: {len(ms)} launches, {tot:.2f} ms inside C-ABI calls")
for lab, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOPN", 70))]:
    print(f"{t:9.3f} ms {n:5d} x {1e3 * t / n:9.1f} us  {lab}")
