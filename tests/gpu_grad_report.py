"""Print, per parameter, the relative L2 error of the HIP backward against torch autograd on the oracle (float64 CPU)."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from util import make_model, rel, synth


def oracle_grads(sd, cfg, mix, emb, wgt, training):
    from oracle.avnet_ref import avnet_forward

    nograd = ("running_mean", "running_var", "scale_x", ".pe")
    sd64 = {}
    for k, v in sd.items():
        if not v.is_floating_point():
            sd64[k] = v.clone()
        else:
            sd64[k] = v.double().clone().requires_grad_(not k.endswith(nograd))
    out = avnet_forward(sd64, cfg, mix.double(), emb.double(), training=training)
    loss = (out * wgt.double()).sum()
    loss.backward()
    return out.detach(), {k: v.grad for k, v in sd64.items() if v.is_floating_point() and v.grad is not None}


def main(B=2, L=4096, R=2, training=False):
    Tv = max(3, 25 * L // 16000)
    model, sd, cfg = make_model(R, "cuda")
    for mod in model.modules():  # the oracle has no dropout: compare with p = 0
        if hasattr(mod, "p") and isinstance(getattr(mod, "p"), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    model.train(training)
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    g = torch.Generator().manual_seed(7)
    wgt = torch.randn(B, 1, L, generator=g)
    out = model(mix.cuda(), emb.cuda())
    (out * wgt.cuda()).sum().backward()
    torch.cuda.synchronize()
    ref_out, ref = oracle_grads(sd, cfg, mix, emb, wgt, training)
    print(f"forward rel={rel(out.detach().cpu(), ref_out):.3e}")
    worst = 0.0
    for n, p in model.named_parameters():
        if n not in ref:
            continue
        if p.grad is None:
            print(f"{n:90s} NO GRAD (ref norm {float(ref[n].norm()):.3e})")
            continue
        e = rel(p.grad.cpu(), ref[n])
        worst = max(worst, e)
        flag = "" if e < 2e-3 else "   <<<<<<"
        print(f"{n:90s} rel={e:.3e} |ref|={float(ref[n].norm()):.3e}{flag}")
    print("worst", worst)


if __name__ == "__main__":
    main(training=len(sys.argv) > 1 and sys.argv[1] == "train")
