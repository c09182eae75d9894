import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from util import make_model, rel, synth
from oracle.avnet_ref import avnet_forward
for R, B, L, Tv in ((2, 1, 8000, 12), (4, 1, 32000, 50), (6, 2, 32000, 50)):
    model, sd, cfg = make_model(R, "cuda")
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
        ref = avnet_forward({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, cfg, mix.double(), emb.double())
    print(R, B, L, "rel L2 vs float64 oracle: %.3e" % rel(out, ref))
