"""GPU micro-benchmark: one VP-block training step (forward + backward) through the HIP kernels vs the PyTorch glue, B = 32, Tv = 50."""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import make_model  # noqa: E402

from rtfs_net_amd.models.vp_train import VPTrainer, vp_block_train  # noqa: E402

model, _, _ = make_model(2, "cuda")
vb = model.refinement_module.video_net.get_block(0).train()
ref = copy.deepcopy(vb)
x = torch.randn(32, 512, 50, device="cuda")
w = torch.randn(32, 512, 50, device="cuda")
tr = VPTrainer(vb)
for name, fn in (("hip", lambda: vp_block_train(tr, x)), ("glue", lambda: ref(x))):
    for _ in range(5):
        (fn() * w).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        (fn() * w).sum().backward()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {1e3 * (t1 - t0) / 30:.2f} ms / step, with drain {1e3 * (t2 - t0) / 30:.2f} ms")
