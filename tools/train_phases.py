"""Wall-clock split of one training step (forward / loss+backward / clip / optimizer), batch 32, RTFS-Net-6."""
import copy
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from oracle import synth  # noqa: E402
from rtfs_net_amd import AVNet  # noqa: E402
from rtfs_net_amd.losses import PITLossWrapper, pairwise_neg_snr  # noqa: E402

B, R, L, Tv = 32, 6, 32000, 50
dev = torch.device("cuda:0")
model = AVNet(print_macs=False, **copy.deepcopy(synth.rtfs_audionet(R)))
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.to(dev).train()
mix, tgt, emb = synth.synth_inputs(B, L, Tv)
mix, tgt, emb = mix.to(dev), tgt.to(dev).unsqueeze(1), emb.to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
loss_fn = PITLossWrapper(pairwise_neg_snr, pit_from="pw_mtx")
acc = [0.0] * 4


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for it in range(6):
    opt.zero_grad(set_to_none=True)
    t0 = sync()
    est = model(mix, emb)
    t1 = sync()
    loss_fn(est, tgt).backward()
    t2 = sync()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    t3 = sync()
    opt.step()
    t4 = sync()
    if it >= 2:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[i] += d * 1e3 / 4
t = sync()
for _ in range(5):
    for p_ in model.parameters():
        p_.data.add_(0)  # bump versions like an optimizer step
    model._trainer.weights()
print("weight re-preparation after an optimizer step: %.1f ms" % ((sync() - t) * 1e3 / 5))
print("forward %.1f ms, loss+backward %.1f ms, clip %.1f ms, optimizer %.1f ms" % tuple(acc))
