"""Bisect a gradient deviation on a tiny input: compares the HIP chain's boundary gradients (d refined, d x0, d a0, d a_emb) with the
float64 oracle's (autograd on the taps).  Usage: python tools/grad_small_bisect.py [L] [R]"""
import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
torch.set_num_threads(16)
from oracle.avnet_ref import avnet_forward
from oracle.regimes import stable_emb
from util import make_model, synth
NOGRAD = ("running_mean", "running_var", "scale_x", ".pe")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2300
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, Tv, training = 3, 17, False
model, sd, cfg = make_model(R, "cuda")
mix, _, emb = synth.synth_inputs(B, L, Tv)
emb = stable_emb(sd, cfg, emb, training)
model.train(training)
wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(7))
import warnings; warnings.simplefilter("ignore")
from rtfs_net_amd.models import hip_train
cap = {}
orig_b, orig_call = hip_train.HipTrainer.backward_b, hip_train.HipTrainer._call
def bb(self, c, dout):
    r = orig_b(self, c, dout)
    cap["dx0"], cap["da0"], cap["da_emb_mask"] = r[0].clone(), (None if r[1] is None else r[1].clone()), r[2].clone()
    cap["shape"] = (c.B, c.T)
    return r
def call(self, name, *a):
    r = orig_call(self, name, *a)
    if name == "rtfs_prelu_bwd" and "dref" not in cap: cap["dref"] = a[3].clone()
    if name == "rtfs_proj_gateway_bwd": cap["da0_final"] = a[7] if a[8] in (0, 1) and a[9] is None else a[9]
    return r
hip_train.HipTrainer.backward_b, hip_train.HipTrainer._call = bb, call
out = model(mix.cuda(), emb.cuda()); (out * wgt.cuda()).sum().backward()
torch.cuda.synchronize()
s = {k: (v.double().clone().requires_grad_(not k.endswith(NOGRAD)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
taps = {}
o64 = avnet_forward(s, cfg, mix.double(), emb.double(), training=training, taps=taps)
for k in ("block0", "caf", "a0", "a_emb") + tuple(f"block{i}" for i in range(1, R)):
    taps[k].retain_grad()
(o64 * wgt.double()).sum().backward()
Bc, T = cap["shape"]
def cl(t): return t.view(Bc, T, 129, -1).permute(0, 3, 1, 2).double().cpu()   # channels-last [B,T,F,C] -> [B,C,T,F]
def rel(a, b): return float((a - b).norm() / b.norm())
last = "caf" if R == 1 else f"block{R-1}"
print("d refined :", rel(cl(cap["dref"]), taps[last].grad))
print("d x0      :", rel(cl(cap["dx0"]), taps["block0"].grad))
if "da0_final" in cap: print("d a0      :", rel(cl(cap["da0_final"]), taps["a0"].grad))
for name, a, b in (("d refined", cl(cap["dref"]), taps[last].grad), ("d x0", cl(cap["dx0"]), taps["block0"].grad)):
    d = (a - b).abs().flatten()
    top = torch.topk(d, 6)
    tot = float((a - b).norm())
    print(name, "norm of oracle", float(b.norm()), "diff norm", tot, "share of the top 6 elements:", [round(float(v) / tot, 3) for v in top.values])
    for v, i in zip(top.values, top.indices):
        idx = [int(j) for j in torch.unravel_index(i, a.shape)]
        print("   ", idx, "hip", float(a.flatten()[i]), "oracle", float(b.flatten()[i]))
