"""Generate the GRADIENT fixtures under tests/golden/ by running THE REFERENCE ITSELF under float64 autograd.

Run once in the authoring container (the reference's Python never travels to the GPU box):

    python -m oracle.gen_golden_grads [--only SUBSTRING]

For every case of oracle/regimes.py (GRAD_CASES on the ordinary synthetic weights = kind "plain"; SMOOTH_CASES on the
smooth-regime weights = kind "smooth", what the split-bf16 step is checked on) it

  1. prepares the inputs exactly as tests/test_hip_backward.py used to on the GPU box: synthetic mixture + lip embeddings
     (oracle/synth.py), kink-stable embeddings (regimes.stable_emb), for "smooth" the smooth-regime state dict
     (regimes.smooth_regime: four float64 forwards of the oracle);
  2. imports `src.models.AVNet` from /root/reference (import stubs of oracle/stubs; sru -> oracle/sru_ref.py), loads the state
     dict, casts to float64, switches dropout / DropPath off (the product test does the same: both sides must see one function),
     runs forward + `(out * wgt).sum().backward()` in eval or train mode (train: BatchNorm batch statistics, running-statistics
     update with momentum 0.1) - this is autograd over /root/reference/src/models/separators/tdanet.py:106-133 and the rest of
     AVNet.forward, the thing the HIP adjoint chain replaces;
  3. runs the same step through float64 autograd of the oracle restatement (oracle/avnet_ref.py) and REFUSES to write the
     fixture if any gradient tensor differs by more than 1e-7 relative (+1e-9 of the largest gradient norm) - which pins the
     oracle's backward to the reference's;
  4. stores data only: emb (float32, the kink-stable embeddings), out (float32 waveform), every parameter gradient as
     float32 (`grad.<key>`), for train mode the 56 running statistics after the step (`stat.<key>`), for "smooth" the state-dict
     entries that differ from the synthetic ones (`sd.<key>`).  float32 storage of a float64 gradient costs 6e-8 relative.

Also writes the VP-block-only step fixtures (tests/test_vp_train.py::test_vp_block_training_step_matches_the_reference) from the
reference's video TDANetBlock (`model.refinement_module.video_net.get_block(0)`) the same way: `vpgrads_<mode>_B<b>_Tv<t>.npz`.
"""
from __future__ import annotations

import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
NOGRAD = ("running_mean", "running_var", "scale_x", ".pe", "num_batches_tracked")

VP_CASES = [(3, 50), (2, 25), (2, 12), (1, 100), (1, 230)]  # (B, Tv) of tests/test_vp_train.py, each in train and eval mode


def _stochastic_layers_off(model):
    for mod in model.modules():
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(getattr(mod, "drop_prob", None), float):
            mod.drop_prob = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0


def _rel(a, b, floor=0.0):
    return float((a - b).norm()) / (float(b.norm()) + floor + 1e-300)


def reference_step(AVNet, cfg, sd, mix, emb, wgt, training):
    """float64 forward + backward of the imported reference; returns (out, {key: grad}, {key: running statistic after the step})"""
    torch.manual_seed(0)
    model = AVNet(print_macs=False, **copy.deepcopy(cfg))
    model.load_state_dict(sd)
    model = model.double()
    _stochastic_layers_off(model)
    model.train(training)
    out = model(mix.double(), emb.double())
    (out * wgt.double()).sum().backward()
    grads = {n: p.grad.detach() for n, p in model.named_parameters() if p.grad is not None}
    stats = {k: v.detach().clone() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    return out.detach(), grads, stats


def oracle_step(cfg, sd, mix, emb, wgt, training):
    from oracle.avnet_ref import avnet_forward

    sd64 = {k: (v.double().clone().requires_grad_(not k.endswith(NOGRAD)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    out = avnet_forward(sd64, cfg, mix.double(), emb.double(), training=training)
    (out * wgt.double()).sum().backward()
    grads = {k: v.grad for k, v in sd64.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
    stats = {k: v.detach() for k, v in sd64.items() if k.endswith(("running_mean", "running_var"))}
    return out.detach(), grads, stats


def full_model_case(AVNet, kind, training, B, L, R, Tv):
    from oracle import regimes, synth

    name = regimes.case_name(kind, training, B, L, R, Tv)
    t0 = time.time()
    cfg = synth.rtfs_audionet(R)
    if kind == "nonshared":
        cfg["audio_params"]["shared"] = False
    torch.manual_seed(0)
    template = AVNet(print_macs=False, **copy.deepcopy(cfg)).state_dict()
    sd0 = synth.synth_state_dict(template)
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    sd = regimes.smooth_regime(sd0, cfg, mix, emb, training) if kind in ("smooth", "nonshared") else sd0  # (nonshared: see regimes.NONSHARED_CASES)
    emb = regimes.stable_emb(sd, cfg, emb, training)
    wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(regimes.GRAD_WEIGHT_SEED))
    out, grads, stats = reference_step(AVNet, cfg, sd, mix, emb, wgt, training)
    o_out, o_grads, o_stats = oracle_step(cfg, sd, mix, emb, wgt, training)
    scale = max(float(g.norm()) for g in grads.values())
    assert set(grads) == set(o_grads), set(grads) ^ set(o_grads)
    worst = max((_rel(o_grads[k], grads[k], 1e-9 * scale), k) for k in grads)
    worst_stat = max((_rel(o_stats[k], stats[k]), k) for k in stats)
    print(f"  {name}: oracle-vs-reference out {_rel(o_out, out):.1e}, worst gradient {worst[0]:.1e} ({worst[1]}), worst running stat {worst_stat[0]:.1e}"
          f"  [{time.time() - t0:.0f} s]", flush=True)
    if _rel(o_out, out) > 1e-9 or worst[0] > 1e-7 or worst_stat[0] > 1e-9:
        raise SystemExit("oracle autograd disagrees with the reference's; fixture NOT written")
    arrays = {"emb": emb.numpy().astype(np.float32), "out": out.numpy().astype(np.float32), "mix_head": mix[:, :256].numpy()}
    arrays.update({f"grad.{k}": v.numpy().astype(np.float32) for k, v in grads.items()})
    if training:
        arrays.update({f"stat.{k}": v.numpy().astype(np.float64) for k, v in stats.items()})
    if kind in ("smooth", "nonshared"):
        arrays.update({f"sd.{k}": v.numpy() for k, v in sd.items() if not torch.equal(v, sd0[k])})
    from oracle.gen_golden import _sru_source

    arrays["sru_source"] = _sru_source()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)


def vp_case(AVNet, train, B, Tv):
    from oracle import avnet_ref, regimes, synth

    cfg = synth.rtfs_audionet(2)
    torch.manual_seed(0)
    model = AVNet(print_macs=False, **copy.deepcopy(cfg))
    sd = synth.synth_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model = model.double()
    _stochastic_layers_off(model)
    vb = model.refinement_module.video_net.get_block(0)
    vb.train(train)
    g = torch.Generator().manual_seed(100 + Tv)
    x = regimes.stable_emb(sd, cfg, torch.randn(B, 512, Tv, generator=g), train)
    wgt = torch.randn(B, 512, Tv, generator=g)
    x64 = x.double().requires_grad_(True)
    out = vb(x64)
    (out * wgt.double()).sum().backward()
    P = regimes.VIDEO_PREFIX
    grads = {n: p.grad.detach() for n, p in vb.named_parameters() if p.grad is not None}
    stats = {n: b.detach().clone() for n, b in vb.named_buffers() if n.endswith(("running_mean", "running_var"))}
    # the oracle's block on the same step
    sd64 = {k: (v.double().clone().requires_grad_(not k.endswith(NOGRAD)) if v.is_floating_point() else v.clone()) for k, v in sd.items() if k.startswith(P)}
    xo = x.double().requires_grad_(True)
    o_out = avnet_ref.tdanet_block(xo, avnet_ref.P(sd64).sub(P), avnet_ref.normalise_cfg(cfg)["video"], training=train)
    (o_out * wgt.double()).sum().backward()
    scale = max(float(v.norm()) for v in grads.values())
    worst = max((_rel(sd64[f"{P}.{k}"].grad, grads[k], 1e-9 * scale), k) for k in grads)
    name = f"vpgrads_{'train' if train else 'eval'}_B{B}_Tv{Tv}"
    print(f"  {name}: oracle-vs-reference out {_rel(o_out.detach(), out.detach()):.1e}, dx {_rel(xo.grad, x64.grad):.1e}, worst gradient {worst[0]:.1e} ({worst[1]})", flush=True)
    if _rel(o_out.detach(), out.detach()) > 1e-9 or worst[0] > 1e-7 or _rel(xo.grad, x64.grad) > 1e-7:
        raise SystemExit("oracle autograd disagrees with the reference's VP block; fixture NOT written")
    arrays = {"x": x.numpy().astype(np.float32), "wgt": wgt.numpy(), "out": out.detach().numpy().astype(np.float32), "dx": x64.grad.numpy().astype(np.float32)}
    arrays.update({f"grad.{k}": v.numpy().astype(np.float32) for k, v in grads.items()})
    if train:
        arrays.update({f"stat.{k}": v.numpy() for k, v in stats.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)


def main():
    from oracle.gen_golden import _import_reference
    from oracle import regimes

    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="substring of the fixture name")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    AVNet = _import_reference()
    torch.set_num_threads(os.cpu_count() or 8)
    for train in (True, False):
        for B, Tv in VP_CASES:
            if args.only in f"vpgrads_{'train' if train else 'eval'}_B{B}_Tv{Tv}":
                vp_case(AVNet, train, B, Tv)
    for kind, cases in (("plain", regimes.GRAD_CASES), ("smooth", regimes.SMOOTH_CASES), ("nonshared", regimes.NONSHARED_CASES)):
        for case in cases:
            if args.only in regimes.case_name(kind, *case):
                full_model_case(AVNet, kind, *case)
    print("done ->", OUT)


if __name__ == "__main__":
    sys.exit(main())
