"""Generate the LONG-utterance fixtures under tests/golden/ by running THE REFERENCE ITSELF (eval mode, float32 as it ships).

    python -m oracle.gen_golden_long [--only SUBSTRING]

The reference has no length limit; the HIP path's long-input forms (1024-key single-tile attention, key-blocked attention past 1024
compressed frames, multi-launch VP chain past Tv = 100, CAF video kernel at hundreds of frames, 32-bit in-utterance offsets up to the
length guard) used to be checked against the oracle's CPU forward ON the GPU box - tens of seconds of host time each.  This script
runs /root/reference/src/models once here instead and stores, per case, a strided sample of the waveform (every STRIDE-th sample),
its full-length norm and the head of the input (to pin the synthetic generator):

    long_8s_R2.npz      L = 136000  (531 compressed frames), Tv = 212
    long_30s_R2.npz     L = 480000  (1875 compressed frames: key-blocked attention), Tv = 750
    long_64s_R1.npz     L = 1024000 (4000 compressed frames), Tv = 1600
    long_120s_R1.npz    L = 1920000 (7500 compressed frames, T = 15001: 1.98 GB per [T][129][256] activation, just inside the 2 GiB guard), Tv = 3000
    scale_x.npz         RTFS-Net-2, B = 2, 1 s with every SRU `scale_x` buffer != 1: waveform + the float64 gradient of one SRU weight

It also checks the oracle restatement against the reference on every case (<= 2e-5) before writing.
"""
from __future__ import annotations

import argparse
import copy
import os
import time

import numpy as np
import torch

from oracle.gen_golden import _sru_source

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
STRIDE = 8
LONG_CASES = [("long_8s_R2", 2, 136000, 212), ("long_30s_R2", 2, 480000, 750), ("long_64s_R1", 1, 1024000, 1600), ("long_120s_R1", 1, 1920000, 3000)]
SCALE_X_GRAD = "refinement_module.audio_net.blocks.globalatt.1.rnn.rnn_lst.2.weight"


def scale_x_state(sd):
    """every one of the 8 SRU layers' `scale_x` buffers set to a different value != 1 (shared with tests/test_hip_fullsize.py)"""
    sd = dict(sd)
    keys = sorted(k for k in sd if k.endswith("scale_x"))
    assert len(keys) == 8
    for j, k in enumerate(keys):
        sd[k] = torch.tensor([0.55 + 0.15 * j])
    return sd


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def long_case(AVNet, name, R, L, Tv, check_oracle=True):
    from oracle import synth
    from oracle.avnet_ref import avnet_forward

    t0 = time.time()
    cfg = synth.rtfs_audionet(R)
    torch.manual_seed(0)
    model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
    sd = synth.synth_state_dict(model.state_dict())
    model.load_state_dict(sd)
    mix, _, emb = synth.synth_inputs(1, L, Tv)
    with torch.no_grad():
        out = model(mix, emb)
        t1 = time.time()
        e = float("nan")
        if check_oracle:
            e = _rel(avnet_forward(sd, cfg, mix, emb), out)
    print(f"  {name}: reference forward {t1 - t0:.0f} s, oracle-vs-reference {e:.2e}", flush=True)
    if check_oracle and e > 2e-5:
        raise SystemExit("oracle disagrees with the reference; fixture NOT written")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), out_strided=out[0, 0, ::STRIDE].numpy().copy(), stride=np.int64(STRIDE),
                        norm=np.float64(out.double().norm()), mix_head=mix[:, :256].numpy(), sru_source=_sru_source())


def scale_x_case(AVNet):
    from oracle import synth
    from oracle.avnet_ref import avnet_forward

    cfg = synth.rtfs_audionet(2)
    torch.manual_seed(0)
    model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
    sd = scale_x_state(synth.synth_state_dict(model.state_dict()))
    model.load_state_dict(sd)
    mix, _, emb = synth.synth_inputs(2, 16000, 25)
    with torch.no_grad():
        out = model(mix, emb)
        sd1 = {k: (torch.ones_like(v) if k.endswith("scale_x") else v) for k, v in sd.items()}
        model.load_state_dict(sd1)
        out1 = model(mix, emb)
        e = _rel(avnet_forward(sd, cfg, mix, emb), out)
    assert _rel(out1, out) > 1e-2  # the buffers matter
    model.load_state_dict(sd)
    model = model.double()
    wgt = torch.randn(2, 1, 16000, generator=torch.Generator().manual_seed(5))
    (model(mix.double(), emb.double()) * wgt.double()).sum().backward()
    grad = dict(model.named_parameters())[SCALE_X_GRAD].grad
    print(f"  scale_x: oracle-vs-reference {e:.2e}, scale_x = 1 moves the waveform by {_rel(out1, out):.2e}", flush=True)
    if e > 2e-5:
        raise SystemExit("oracle disagrees with the reference; fixture NOT written")
    np.savez_compressed(os.path.join(OUT, "scale_x.npz"), out=out.numpy(), grad=grad.numpy().astype(np.float32), mix_head=mix[:, :256].numpy(),
                        sru_source=_sru_source())


def main():
    from oracle.gen_golden import _import_reference

    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    AVNet = _import_reference()
    torch.set_num_threads(os.cpu_count() or 8)
    if args.only in "scale_x":
        scale_x_case(AVNet)
    for name, R, L, Tv in LONG_CASES:
        if args.only in name:
            long_case(AVNet, name, R, L, Tv, check_oracle=L <= 1024000)  # (120 s: the reference alone; the oracle is checked at the shorter lengths)
    print("done ->", OUT)


if __name__ == "__main__":
    main()
