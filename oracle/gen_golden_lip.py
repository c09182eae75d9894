"""Generate tests/golden/lip.npz by RUNNING THE REFERENCE lip encoder (run once in the authoring container):

    python -m oracle.gen_golden_lip

Imports `src.models.videomodels.FRCNNVideoModel` from /root/reference (read-only; thop stubbed by oracle/stubs), loads the
deterministic weights of oracle/synth.py, pushes the synthetic crops of oracle/lip_ref.lip_inputs through it in eval mode and stores
DATA only: the output embeddings and strided samples of the stage boundaries (weights and inputs are regenerated from seeds).
Refuses to write if the oracle restatement (oracle/lip_ref.py) disagrees with the reference by more than 2e-5 relative L2."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CASES = {"a": (2, 5, 88, 88), "b": (1, 3, 96, 96), "c": (1, 2, 45, 51)}  # B, T, H, W  (b: other crop size; c: odd sizes)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.ref_import import prepare_path

    prepare_path()  # /root/reference first, the import stubs last: packages the environment has are used as they are
    from src.models.videomodels import FRCNNVideoModel  # the reference

    from oracle import synth
    from oracle.lip_ref import frcnn_forward, lip_inputs

    torch.manual_seed(0)
    model = FRCNNVideoModel(print_macs=False)
    model.eval()  # (the reference's train() override returns None, so eval() cannot be chained)
    sd = synth.synth_state_dict(model.state_dict(), salt=3)
    model.load_state_dict(sd)
    out = {}
    for name, (B, T, H, W) in CASES.items():
        x = lip_inputs(B, T, H, W)
        taps_ref = {}
        hooks = [model.frontend3D.register_forward_hook(lambda m, i, o: taps_ref.__setitem__("front", o.transpose(1, 2).reshape(-1, o.shape[1], o.shape[3], o.shape[4])))]
        for li in range(1, 5):
            hooks.append(getattr(model.trunk, f"layer{li}").register_forward_hook(lambda m, i, o, li=li: taps_ref.__setitem__(f"layer{li}", o)))
        with torch.no_grad():
            y_ref = model(x)
        for h in hooks:
            h.remove()
        taps = {}
        y_or = frcnn_forward(sd, x, taps)
        e = rel(y_or, y_ref)
        print(f"case {name}: B={B} T={T} {H}x{W}  out {tuple(y_ref.shape)}  oracle vs reference rel {e:.2e}")
        assert e < 2e-5, e
        for k in taps_ref:
            assert rel(taps[k], taps_ref[k]) < 2e-5, (k, rel(taps[k], taps_ref[k]))
        y64 = frcnn_forward({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, x.double())
        print(f"         float64 oracle vs reference rel {rel(y64.float(), y_ref):.2e}")
        out[f"{name}_shape"] = np.array([B, T, H, W])
        out[f"{name}_out"] = y_ref.numpy()
        for k, v in taps_ref.items():
            out[f"{name}_{k}"] = v[:, ::7, ::3, ::3].contiguous().numpy()
    out["keys"] = np.array(sorted(f"{k}:{tuple(v.shape)}" for k, v in sd.items()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lip.npz"), **out)
    print("wrote tests/golden/lip.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "lip.npz")), "bytes")


if __name__ == "__main__":
    main()
