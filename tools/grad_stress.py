"""GPU: the training-step path of a small case (eval mode under autograd, B = 2, 4096 samples, RTFS-Net-2, 6 video frames - the first case of
tests/test_hip_backward.py) N times in one process; every parameter gradient of every run against the first run's (the sums use atomics: 1e-3 per tensor).
A run that differs is a race somewhere in the step.  python tools/grad_stress.py [N] [R] [L]"""
import sys

import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from util import make_model, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
model, _, _ = make_model(R, "cuda")
model.eval()
mix, _, emb = synth.synth_inputs(2, L, 6)
mix, emb = mix.cuda(), emb.cuda()
wgt = torch.randn(2, 1, L, generator=torch.Generator().manual_seed(1)).cuda()
ref, bad = None, 0
for it in range(N):
    model.zero_grad(set_to_none=True)
    (model(mix, emb) * wgt).sum().backward()
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    if ref is None:
        ref = g
        scale = max(float(v.norm()) for v in ref.values())
        continue
    for n, v in g.items():
        err = float((v - ref[n]).norm()) / (float(ref[n].norm()) + 1e-4 * scale)
        if err > 1e-3:
            bad += 1
            print(f"run {it}: {n} differs from run 0 by {err:.3e} (norm {float(v.norm()):.3e} against {float(ref[n].norm()):.3e})")
print(f"{N} runs, {bad} differing tensors")
