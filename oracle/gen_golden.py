"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run once in the authoring container (the reference's Python never travels to the GPU box):

    python -m oracle.gen_golden

It imports `src.models.AVNet` from /root/reference (read-only); third-party packages the
reference needs come from the environment when it has them and from the import stubs of
oracle/stubs otherwise (sru -> oracle/sru_ref.py, timm DropPath, thop, pytorch_lightning:
oracle/ref_import.py; every fixture records which SRU ran under `sru_source`), loads the
deterministic weights of oracle/synth.py, pushes the synthetic inputs through it in eval mode and
stores inputs/outputs -- data only, no reference source:

  tiny.npz        reduced RTFS-Net (oracle.synth.TINY_AUDIONET), B=2, L=1024, Tv=13:
                  inputs, output waveform and every stage boundary captured with forward hooks
                  (encoder, bottleneck, RTFS block internals, VP block, CAF, every block, mask).
  tiny_state.npz  the weights used for it (reference key names) -- also pins the key/shape list.
  rtfs4_b1.npz / rtfs6_b2.npz / rtfs12_4s_b1.npz
                  full-size RTFS-Net-4/6 (2 s) and RTFS-Net-12 (4 s): output waveform + strided
                  samples of stage boundaries; weights are regenerated from oracle/synth.py.
  state_keys.json key -> shape of the full-size reference state dict.

While doing so it checks the oracle restatement (oracle/avnet_ref.py) against the reference and
refuses to write fixtures if they disagree by more than 2e-5 relative L2.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    """the reference's AVNet; a real `sru` package in the environment wins over the stub (oracle/ref_import.py)"""
    from oracle.ref_import import import_reference, sru_source

    AVNet = import_reference()
    print("SRU arithmetic of this run:", sru_source())
    return AVNet


def _sru_source():
    from oracle.ref_import import sru_source

    return np.array(sru_source())


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def run_reference(AVNet, audionet, B, L, Tv, salt=0):
    import copy

    from oracle import synth
    from oracle.avnet_ref import avnet_forward

    torch.manual_seed(0)
    model = AVNet(print_macs=False, **copy.deepcopy(audionet)).eval()
    sd = synth.synth_state_dict(model.state_dict(), salt)
    model.load_state_dict(sd)
    mix, s1, emb = synth.synth_inputs(B, L, Tv, audionet["pretrained_vout_chan"])

    taps = {}

    def hook(name):
        def f(mod, inp, out):
            n, i = name, 0
            while n in taps:  # shared blocks are called several times
                i += 1
                n = f"{name}#{i}"
            taps[n] = out.detach().clone()

        return f

    rm = model.refinement_module
    ab = rm.audio_net.blocks
    hooks = [
        model.encoder.register_forward_hook(hook("a_emb")),
        model.audio_bottleneck.register_forward_hook(hook("a0")),
        ab.register_forward_hook(hook("block")),
        ab.gateway.register_forward_hook(hook("gateway")),
        ab.projection.register_forward_hook(hook("projection")),
        ab.downsample_layers[0].register_forward_hook(hook("down0")),
        ab.downsample_layers[1].register_forward_hook(hook("down1")),
        ab.globalatt[0].register_forward_hook(hook("dp_freq")),
        ab.globalatt[1].register_forward_hook(hook("dp_time")),
        ab.globalatt[2].register_forward_hook(hook("attn")),
        ab.fusion_layers[0].register_forward_hook(hook("tfar0")),
        ab.fusion_layers[1].register_forward_hook(hook("tfar1")),
        ab.concat_layers[0].register_forward_hook(hook("concat0")),
        rm.video_net.blocks.register_forward_hook(hook("vp")),
        rm.crossmodal_fusion.fusion_module.audio_lstm.register_forward_hook(hook("caf")),
        model.mask_generator.register_forward_hook(hook("masked")),
    ]
    with torch.no_grad():
        out = model(mix, emb)
    for h in hooks:
        h.remove()

    # oracle vs reference (this is what pins oracle/avnet_ref.py)
    otaps = {}
    with torch.no_grad():
        o = avnet_forward(sd, audionet, mix, emb, taps=otaps)
    errs = {"out": rel(o, out), "a_emb": rel(otaps["a_emb"], taps["a_emb"]), "a0": rel(otaps["a0"], taps["a0"]),
            "block0": rel(otaps["block0"], taps["block"]), "caf": rel(otaps["caf"], taps["caf"]),
            "vp": rel(otaps["vp"], taps["vp"]), "masked": rel(otaps["masked"], taps["masked"]),
            "pooled->freq": rel(otaps["block0.globalatt.0"], taps["dp_freq"]),
            "time": rel(otaps["block0.globalatt.1"], taps["dp_time"]),
            "attn": rel(otaps["block0.globalatt.2"], taps["attn"])}
    worst = max(errs.values())
    print("  oracle-vs-reference rel L2:", {k: f"{v:.2e}" for k, v in errs.items()})
    if worst > 2e-5:
        raise SystemExit(f"oracle disagrees with the reference ({worst:.3e}); fixtures NOT written")
    return model, sd, mix, s1, emb, out, taps


def strided(t: torch.Tensor, n=4096):
    f = t.flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def main():
    os.makedirs(OUT, exist_ok=True)
    AVNet = _import_reference()
    from oracle import synth

    print("tiny config")
    model, sd, mix, s1, emb, out, taps = run_reference(AVNet, synth.TINY_AUDIONET, B=2, L=1024, Tv=13)
    np.savez_compressed(os.path.join(OUT, "tiny_state.npz"), **{k: v.numpy() for k, v in sd.items()})
    np.savez_compressed(
        os.path.join(OUT, "tiny.npz"), mix=mix.numpy(), s1=s1.numpy(), emb=emb.numpy(), out=out.numpy(), sru_source=_sru_source(),
        **{f"tap.{k}": v.numpy() for k, v in taps.items()},
    )

    keys = None
    for name, R, B, L in (("rtfs4_b1", 4, 1, 32000), ("rtfs6_b2", 6, 2, 32000), ("rtfs12_4s_b1", 12, 1, 64000)):
        print(name)
        Tv = 25 * L // 16000
        cfg = synth.rtfs_audionet(R)
        model, sd, mix, s1, emb, out, taps = run_reference(AVNet, cfg, B=B, L=L, Tv=Tv)
        if keys is None:
            keys = {k: list(v.shape) for k, v in sd.items()}
            with open(os.path.join(OUT, "state_keys.json"), "w") as f:
                json.dump(keys, f, indent=0)
        np.savez_compressed(
            os.path.join(OUT, f"{name}.npz"), out=out.numpy().astype(np.float32), mix_head=mix[:, :256].numpy(), sru_source=_sru_source(),
            **{f"tap.{k}": strided(v) for k, v in taps.items()},
            **{f"norm.{k}": np.float64(v.double().norm()) for k, v in taps.items()},
        )
    print("done ->", OUT)


if __name__ == "__main__":
    main()
