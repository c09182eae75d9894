"""GPU diagnostic: per parameter, error of the HIP gradient AND of torch-fp32 autograd of the oracle against float64 autograd."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hip_backward import _oracle_grads  # noqa: E402
from util import make_model, synth  # noqa: E402


def main(B=1, L=32000, R=2, Tv=50, seed=synth.INPUT_SEED, top=40):
    model, sd, cfg = make_model(R, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, Tv, seed=seed)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(7))
    out = model(mix.cuda(), emb.cuda())
    (out * wgt.cuda()).sum().backward()
    _, ref, _ = _oracle_grads(sd, cfg, mix, emb, wgt, False)
    _, g32, _ = _oracle_grads(sd, cfg, mix, emb, wgt, False, torch.float32)
    scale = max(float(g.norm()) for g in ref.values())
    rows = []
    for n, p in model.named_parameters():
        if n not in ref:
            continue
        den = float(ref[n].norm()) + 1e-4 * scale
        e_hip = float((p.grad.double().cpu() - ref[n]).norm()) / den
        e_32 = float((g32[n].double() - ref[n]).norm()) / den
        e_h32 = float((p.grad.double().cpu() - g32[n].double()).norm()) / den
        rows.append((e_hip / max(e_32, 1e-7), e_hip, e_32, e_h32, n))
    rows.sort(reverse=True)
    print("ratio   hip-vs-64  fp32-vs-64 hip-vs-fp32  name")
    print("seed", seed, "median hip-vs-64 %.3e  median fp32-vs-64 %.3e" % (sorted(r[1] for r in rows)[len(rows) // 2], sorted(r[2] for r in rows)[len(rows) // 2]))
    for r in rows[:top]:
        print("%7.1f %.3e %.3e %.3e %s" % r)


if __name__ == "__main__":
    for sd_ in (int(a) for a in sys.argv[2:]) if len(sys.argv) > 2 else (synth.INPUT_SEED,):
        main(seed=sd_, top=int(sys.argv[1]) if len(sys.argv) > 1 else 40)
