"""GPU: which entry point makes the plain-bf16 forward differ from run to run?  Runs the forward in bf16 mode with ONE entry point (or all but
one) switched to the split-bf16 product (terms 3: deterministic, same packed weights) and counts the runs that differ from the first.
usage: python tools/bf16_bisect.py [R] [B] [L] [Tv] [runs]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import AVNet, lib, synthetic  # noqa: E402
import rtfs_net_amd.models.hip_path as hp  # noqa: E402

R, B, L, Tv, runs = [int(a) for a in (sys.argv[1:6] + ["6", "32", "32000", "50", "40"][len(sys.argv) - 1:])]
cfg = synthetic.rtfs_audionet(R)
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synthetic.synth_state_dict(model.state_dict()))
model = model.cuda()
mix, _, emb = synthetic.synth_inputs(B, L, Tv)
mix, emb = mix.cuda(), emb.cuda()
model.set_compute_dtype("bf16")
names = ["rtfs_attn_core_fwd", "rtfs_attn_out_fwd", "rtfs_attn_qkv_fwd", "rtfs_bottleneck_fwd", "rtfs_dp_convt_fwd", "rtfs_dp_unfold_gemm_fwd",
         "rtfs_gemm_rows_fwd", "rtfs_mask_fwd", "rtfs_proj_fwd", "rtfs_resid_caf_fwd", "rtfs_resid_fwd", "rtfs_resid_proj_fwd", "rtfs_sru_layer_fwd"]
force3 = set()
orig = hp.HipPath._mm if hasattr(hp, "HipPath") else None
cls = type(model._hip)


def mm(self, name, *args):
    lib.call(name + "_bf16", *args, 3 if name in force3 else self.prec)


cls._mm = mm


def count():
    with torch.no_grad():
        ref = model(mix, emb)
        bad = 0
        for _ in range(runs):
            bad += int(not torch.equal(model(mix, emb), ref))
    return bad


print("all bf16:", count(), "of", runs, flush=True)
force3 = set(names)
print("all split-bf16 through the override:", count(), "of", runs, flush=True)
for n in names:
    force3 = {n}
    a = count()
    force3 = set(names) - {n}
    b = count()
    print(f"{n}: only this one split -> {a} of {runs} differ;  only this one plain bf16 -> {b} of {runs} differ", flush=True)
