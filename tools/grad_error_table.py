"""Per-parameter error of the HIP training step against the float64 fixtures (tests/golden/grads_*.npz), sorted: where does the step lose
precision?  (The oracle's own fp32 autograd holds 2e-5 on every tensor of these cases: tools/slope_grad_probe.py.)
    python tools/grad_error_table.py [case indices ...] [--dtype f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle.regimes import GRAD_CASES, GRAD_WEIGHT_SEED, case_name
from util import load_npz, make_model, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
dtype = "f32"
for case in [GRAD_CASES[int(i)] for i in (args or range(len(GRAD_CASES)))]:
    training, B, L, R, Tv = case
    model, sd, cfg = make_model(R, "cuda")
    for mod in model.modules():
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    z = load_npz(case_name("plain", *case) + ".npz")
    mix, _, _ = synth.synth_inputs(B, L, Tv)
    emb = torch.from_numpy(z["emb"])
    model.train(training)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(GRAD_WEIGHT_SEED))
    out = model(mix.cuda(), emb.cuda())
    (out * wgt.cuda()).sum().backward()
    ref = {k[5:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("grad.")}
    scale = max(float(g.norm()) for g in ref.values())
    rows = []
    for n, p in model.named_parameters():
        if float(ref[n].norm()) < 1e-6 * scale:
            continue
        e = float((p.grad.double().cpu() - ref[n]).norm()) / (float(ref[n].norm()) + 1e-4 * scale)
        rows.append((e, n, p.numel(), float(ref[n].norm()) / scale))
    rows.sort(reverse=True)
    print(f"== {case}: {len(rows)} tensors, median {rows[len(rows)//2][0]:.1e}")
    for e, n, k, rn in rows[:40]:
        print(f"   {e:.2e}  numel {k:7d}  |ref|/max {rn:.1e}  {n}")
    print("   -- scalar parameters (numel <= 12):")
    for e, n, k, rn in rows:
        if k <= 12 and e > 2e-4:
            print(f"   {e:.2e}  numel {k:3d}  |ref|/max {rn:.1e}  {n}")
