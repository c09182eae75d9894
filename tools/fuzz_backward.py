"""Random-shape fuzz of the training step's parameter gradients against float64 autograd of the oracle (fixed seed; kink-stable lip embeddings as in
the gradient tests; prints the median / worst per-tensor error per case - audio-branch activation kinks can move single tensors on unlucky inputs)."""
import random
import sys

import torch

sys.path.insert(0, "/root/repo/tests")
sys.path.insert(0, "/root/repo")
torch.set_num_threads(16)
from oracle.avnet_ref import avnet_forward  # noqa: E402
from oracle.regimes import stable_emb  # noqa: E402
from util import make_model, synth  # noqa: E402

random.seed(2)
NOGRAD = ("running_mean", "running_var", "scale_x", ".pe")
for it in range(10):
    R, B = random.choice([1, 2, 3]), random.choice([1, 2, 3])
    L, Tv, training = random.randint(1920, 9000), random.randint(8, 22), random.random() < 0.5
    model, sd, cfg = make_model(R, "cuda")
    for mod in model.modules():
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    emb = stable_emb(sd, cfg, emb, training)
    model.train(training)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(7))
    out = model(mix.cuda(), emb.cuda())
    (out * wgt.cuda()).sum().backward()
    sd64 = {k: (v.double().clone().requires_grad_(not k.endswith(NOGRAD)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    (avnet_forward(sd64, cfg, mix.double(), emb.double(), training=training) * wgt.double()).sum().backward()
    ref = {k: v.grad for k, v in sd64.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
    scale = max(float(g.norm()) for g in ref.values())
    errs = []
    for n, p in model.named_parameters():
        if float(ref[n].norm()) < 1e-6 * scale:
            continue
        errs.append((float((p.grad.double().cpu() - ref[n]).norm()) / (float(ref[n].norm()) + 1e-4 * scale), n))
    errs.sort()
    print(f"train={training} R {R} B {B} L {L} Tv {Tv}: median {errs[len(errs) // 2][0]:.1e}, worst {errs[-1][0]:.1e} ({errs[-1][1][-60:]}), > 3e-3: {sum(e > 3e-3 for e, _ in errs)} of {len(errs)}", flush=True)
