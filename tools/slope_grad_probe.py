"""Is the 4e-3 deviation of the 15 scalar PReLU-slope gradients a property of fp32 SUMMATION in the HIP reducers, or of fp32 arithmetic in the
chain before them?  CPU probe: the oracle's own fp32 autograd (torch CPU sums pairwise / in vector lanes: its summation error is ~1e-7) against the
float64 fixtures of tests/golden/grads_*.npz.  If the oracle's fp32 slopes are off by the same 1e-3 ... 5e-3, fp64 accumulators cannot fix it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth
from oracle.avnet_ref import avnet_forward
from oracle.regimes import GRAD_CASES, GRAD_WEIGHT_SEED, case_name

torch.set_num_threads(8)
for case in [GRAD_CASES[i] for i in (int(a) for a in (sys.argv[1:] or ["5", "0"]))]:
    training, B, L, R, Tv = case
    z = np.load(os.path.join("tests/golden", case_name("plain", *case) + ".npz"))
    cfg = synth.rtfs_audionet(R)
    from rtfs_net_amd import AVNet
    import copy
    sd = synth.synth_state_dict(AVNet(print_macs=False, **copy.deepcopy(cfg)).state_dict())
    mix, _, _ = synth.synth_inputs(B, L, Tv)
    emb = torch.from_numpy(z["emb"])
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(GRAD_WEIGHT_SEED))
    nograd = ("running_mean", "running_var", "scale_x", ".pe", "num_batches_tracked")
    sd32 = {k: (v.float().clone().requires_grad_(not k.endswith(nograd)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    (avnet_forward(sd32, cfg, mix, emb, training=training) * wgt).sum().backward()
    ref = {k[5:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("grad.")}
    scale = max(float(g.norm()) for g in ref.values())
    es, et = [], []
    for n, r in ref.items():
        g = sd32[n].grad
        if g is None or float(r.norm()) < 1e-6 * scale:
            continue
        e = float((g.double() - r).norm()) / (float(r.norm()) + 1e-4 * scale)
        (es if r.numel() <= 12 else et).append((e, n))
    es.sort(reverse=True); et.sort(reverse=True)
    print(case, "oracle fp32 autograd vs float64 fixture: scalar slopes worst", [(f"{e:.1e}", n[-60:]) for e, n in es[:5]], "| tensors worst", f"{et[0][0]:.1e}", "median", f"{et[len(et)//2][0]:.1e}")
