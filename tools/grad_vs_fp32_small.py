import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
torch.set_num_threads(16)
from oracle.avnet_ref import avnet_forward
from oracle.regimes import stable_emb
from util import make_model, synth
NOGRAD = ("running_mean", "running_var", "scale_x", ".pe")
def ograds(sd, cfg, mix, emb, wgt, training, dt):
    s = {k: (v.to(dt).clone().requires_grad_(not k.endswith(NOGRAD)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    (avnet_forward(s, cfg, mix.to(dt), emb.to(dt), training=training) * wgt.to(dt)).sum().backward()
    return {k: v.grad.double() for k, v in s.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
for (R,B,L,Tv,training) in ((1,3,2212,17,False),(1,3,2300,17,False),(1,3,2250,17,False)):
    model, sd, cfg = make_model(R, "cuda")
    for mod in model.modules():
        if isinstance(getattr(mod, "p", None), float): mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    emb = stable_emb(sd, cfg, emb, training)
    model.train(training)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(7))
    out = model(mix.cuda(), emb.cuda()); (out * wgt.cuda()).sum().backward()
    r64, r32 = ograds(sd, cfg, mix, emb, wgt, training, torch.float64), ograds(sd, cfg, mix, emb, wgt, training, torch.float32)
    scale = max(float(g.norm()) for g in r64.values())
    eh, eo = [], []
    for n, p in model.named_parameters():
        if float(r64[n].norm()) < 1e-6 * scale: continue
        d = float(r64[n].norm()) + 1e-4 * scale
        eh.append(float((p.grad.double().cpu() - r64[n]).norm()) / d); eo.append(float((r32[n] - r64[n]).norm()) / d)
    eh.sort(); eo.sort()
    print(f"R {R} B {B} L {L} train={training}: HIP vs fp64 median {eh[len(eh)//2]:.1e} worst {eh[-1]:.1e};  oracle-fp32 vs fp64 median {eo[len(eo)//2]:.1e} worst {eo[-1]:.1e}")
