"""Boundary gradients of the non-shared-blocks step (HIP chain vs float64 oracle): d refined, d(block 1 input) = d caf, d x0 (block 0 output), d a0."""
import sys, copy, warnings, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
warnings.simplefilter("ignore")
torch.set_num_threads(16)
from oracle.avnet_ref import avnet_forward
from oracle.regimes import NONSHARED_CASES, GRAD_WEIGHT_SEED, case_name
from util import load_npz, synth
from rtfs_net_amd import AVNet
from rtfs_net_amd.models import hip_train
training, B, L, R, Tv = NONSHARED_CASES[0]
shared = len(sys.argv) > 1 and sys.argv[1] == "shared"
cfg = synth.rtfs_audionet(R); cfg["audio_params"]["shared"] = shared
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
sd = synth.synth_state_dict(model.state_dict()); model.load_state_dict(sd); model = model.cuda()
z = load_npz(case_name("nonshared", training, B, L, R, Tv) + ".npz")
mix, _, _ = synth.synth_inputs(B, L, Tv); emb = torch.from_numpy(z["emb"])
wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(GRAD_WEIGHT_SEED))
cap = {}
orig_b, orig_blk = hip_train.HipTrainer._backward_b, hip_train.HipTrainer._block_bwd
def bb(self, c, dout):
    r = orig_b(self, c, dout)
    cap["dx0"], cap["shape"] = r[0].clone(), (c.B, c.T)
    return r
def blk(self, dx, k, bw, B_, T, T2, gr, da0, a0_mode, tag="blk."):
    cap.setdefault("din", []).append((tag, a0_mode, dx.clone()))
    r = orig_blk(self, dx, k, bw, B_, T, T2, gr, da0, a0_mode, tag=tag)
    cap.setdefault("dout", []).append((tag, a0_mode, r.clone()))
    return r
hip_train.HipTrainer._backward_b, hip_train.HipTrainer._block_bwd = bb, blk
out = model(mix.cuda(), emb.cuda()); (out * wgt.cuda()).sum().backward(); torch.cuda.synchronize()
NOGRAD = ("running_mean", "running_var", "scale_x", ".pe")
s = {k: (v.double().clone().requires_grad_(not k.endswith(NOGRAD)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
taps = {}
o64 = avnet_forward(s, cfg, mix.double(), emb.double(), training=training, taps=taps)
for k in ("block0", "caf", "a0", "block1"): taps[k].retain_grad()
(o64 * wgt.double()).sum().backward()
Bc, T = cap["shape"]
def cl(t): return t.view(Bc, T, 129, -1).permute(0, 3, 1, 2).double().cpu()
def rel(a, b): return float((a - b).norm() / b.norm())
print("forward", rel(out.detach().double().cpu(), o64.detach()))
for (tag, mode, t) in cap["din"]:
    ref = taps["block1"].grad if mode in (1, 2) else taps["block0"].grad
    print("block input-side gradient (d block output)", tag, "a0_mode", mode, rel(cl(t), ref))
for (tag, mode, t) in cap["dout"]:
    ref = taps["caf"].grad if mode in (1, 2) else taps["a0"].grad
    print("block result (d block input / running d a0)", tag, "a0_mode", mode, rel(cl(t), ref))
print("d x0", rel(cl(cap["dx0"]), taps["block0"].grad))
tag, mode, t = cap["din"][0]
a_, b_ = cl(t), taps["block1"].grad
d = (a_ - b_).flatten(); tot = float(d.norm())
top = torch.topk(d.abs(), 10)
print("d refined: diff norm", tot, "of", float(b_.norm()), "share of top 1 / 10 elements", round(float(top.values[0]) / tot, 4), round(float((top.values.double() ** 2).sum().sqrt()) / tot, 4))
i = int(top.indices[0]); idx = [int(j) for j in torch.unravel_index(torch.tensor(i), a_.shape)]
print("   element", idx, "hip", float(a_.flatten()[i]), "oracle", float(b_.flatten()[i]), "| oracle refined value there", float(taps["block1"].detach()[tuple(idx)]))
