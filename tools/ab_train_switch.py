"""Same-box, same-process alternating A/B (a, b, b, a, ...) of a host-side switch of the TRAINING step at the config-3 shape (RTFS-Net-6, batch 32, 2 s; the step of
bench.py --mode train: forward + backward + fused AdamW):   python tools/ab_train_switch.py dwadj      (HipForward.fuse entry: 1 = on, 0 = off)"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import AVNet, synthetic as synth  # noqa: E402
from rtfs_net_amd.losses import PITLossWrapper, pairwise_neg_snr  # noqa: E402
from rtfs_net_amd.optim import FusedAdamW  # noqa: E402

name = sys.argv[1]
B = int(os.environ.get("AB_BATCH", "32"))
cfg = synth.rtfs_audionet(6)
model = AVNet(print_macs=False, **copy.deepcopy(cfg))
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.cuda().train()
mix, s1, emb = synth.synth_inputs(B, 32000, 50)
mix, tgt, emb = mix.cuda(), s1.cuda().unsqueeze(1), emb.cuda()
opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
loss_fn = PITLossWrapper(pairwise_neg_snr, pit_from="pw_mtx")


def step():
    opt.zero_grad(set_to_none=True)
    loss_fn(model(mix, emb), tgt).backward()
    opt.step(max_norm=5.0)


def run(v, n=12):
    model._hip.fuse[name] = bool(v)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


res = {0: [], 1: []}
for rep in range(3):
    for v in (0, 1, 1, 0):
        ms = run(v)
        res[v].append(ms)
        print(f"  {name}={v}: median {ms:.2f} ms", flush=True)
for v in (0, 1):
    r = sorted(res[v])
    print(f"{name}={v}: median of medians {r[len(r) // 2]:.2f} ms  (min {r[0]:.2f}, max {r[-1]:.2f})")
