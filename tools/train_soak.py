"""Soak of the training step (RTFS-Net-6, batch 32, 2 s): N steps of forward + backward + AdamW with the weight-gradient side stream on; prints the step time
and the allocator's peak / current bytes every 10 steps - a leak or an allocator that cannot reuse side-stream blocks shows as growth."""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from util import make_model, synth  # noqa: E402
from rtfs_net_amd.losses import PITLossWrapper, pairwise_neg_snr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
model, _, _ = make_model(6, "cuda")
model.train()
mix, tgt, emb = synth.synth_inputs(32, 32000, 50)
mix, tgt, emb = mix.cuda(), tgt.cuda().unsqueeze(1), emb.cuda()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
loss_fn = PITLossWrapper(pairwise_neg_snr, pit_from="pw_mtx")
t0 = time.time()
for it in range(n):
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(model(mix, emb), tgt)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    opt.step()
    if it % 10 == 9:
        torch.cuda.synchronize()
        print(f"step {it + 1}: loss {float(loss):.4f}  {1e3 * (time.time() - t0) / 10:.1f} ms/step  allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
        t0 = time.time()
