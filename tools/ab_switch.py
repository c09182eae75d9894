"""Same-box, same-process A/B of a HOST-side switch of the inference path at the headline shape (RTFS-Net-6, batch 32, 2 s): the settings are
alternated (a, b, b, a, ...) so that clock / temperature drift cancels.

    python tools/ab_switch.py sub_batch 0 8 4        # HipForward attribute, values to compare (first = baseline)
    python tools/ab_switch.py fuse.decmask 1 0        # an entry of HipForward.fuse
"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import AVNet, synthetic as synth  # noqa: E402

name, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
B = int(os.environ.get("AB_BATCH", "32"))
cfg = synth.rtfs_audionet(int(os.environ.get("AB_LAYERS", "6")))
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.cuda()
mix, _, emb = synth.synth_inputs(B, 32000, 50)
mix, emb = mix.cuda(), emb.cuda()


def setv(v):
    if name.startswith("fuse."):
        model._hip.fuse[name[5:]] = bool(v)
    else:
        setattr(model._hip, name, v)


def run(v, n=20):
    setv(v)
    with torch.no_grad():
        for _ in range(3):
            out = model(mix, emb)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record()
            out = model(mix, emb)
            b.record()
        torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], float(out.double().abs().sum())


res = {v: [] for v in values}
order = values + values[::-1]
for rep in range(3):
    for v in order:
        ms, chk = run(v)
        res[v].append(ms)
        print(f"  {name}={v}: median {ms:.3f} ms  checksum {chk:.8e}", flush=True)
for v in values:
    r = sorted(res[v])
    print(f"{name}={v}: median of medians {r[len(r) // 2]:.3f} ms  (min {r[0]:.3f}, max {r[-1]:.3f})")
