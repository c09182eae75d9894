"""Same-box A/B of the whole forward at several batch sizes for the library named by RTFS_HIP_LIB (older builds lack newer entry points: their
declarations are dropped before loading).    RTFS_HIP_LIB=exp/r3/librtfs_hip.so python tools/ab_forward.py [layers]"""
import copy
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib, synthetic as synth  # noqa: E402

raw = ctypes.CDLL(lib.library_path())
for name in list(lib.SIGNATURES):
    if not hasattr(raw, name):
        print("  (library lacks", name + ")")
        lib.SIGNATURES.pop(name)
from rtfs_net_amd import AVNet  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = synth.rtfs_audionet(R)
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.cuda()
print(lib.library_path())
for B in (1, 2, 4, 8, 16, 32):
    mix, _, emb = synth.synth_inputs(B, 32000, 50)
    mix, emb = mix.cuda(), emb.cuda()
    with torch.no_grad():
        for _ in range(5):
            out = model(mix, emb)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record()
            out = model(mix, emb)
            b.record()
        torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"  B {B:2d}: median {t[15]:.3f} ms  min {t[0]:.3f} ms   checksum {float(out.double().abs().sum()):.8e}")

# per-entry-point time at small batch (HIP events around every C-ABI launch; adds ~10 us of event overhead per launch to the wall time, not to the events)
for B in (1, 8):
    mix, _, emb = synth.synth_inputs(B, 32000, 50)
    mix, emb = mix.cuda(), emb.cuda()
    agg = {}
    with torch.no_grad():
        for _ in range(3):
            model(mix, emb)
        for _ in range(5):
            lib.profile_begin("*")
            model(mix, emb)
            torch.cuda.synchronize()
            durs = lib.profile_end()
            for lab, d in zip(lib.profile_labels(), durs):
                k = lab.split("(")[0]
                agg[k] = agg.get(k, 0.0) + d / 5
    print(f"  B {B}: per entry point (ms per forward): " + ", ".join(f"{k[5:]} {v:.3f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]))
