"""CPU: host-side logic added in round 3 (no kernel runs): the dropout / DropPath mask draw of the VP block's HIP training step, the
kink-stability helpers of the gradient tests, bench.py's launcher path."""
import os
import subprocess
import sys

import torch

from oracle.regimes import _video_margin, smooth_regime, stable_emb
from util import make_model, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_attention_mask_draw_layout_and_scaling():
    from rtfs_net_amd.models import vp_train as vt

    model, _, _ = make_model(2)
    ga = model.refinement_module.video_net.get_block(0).globalatt[0]
    assert vt.attn_supported(ga) and len(vt.attn_params(ga)) == 16
    assert sum(p.numel() for p in vt.attn_params(ga)) == 34176  # = VaOff::total of csrc/vp_attn.hip
    ga.eval()
    assert vt.attn_masks(ga, 4, 7, torch.device("cpu")) is None  # eval: no stochastic layer
    ga.train()
    torch.manual_seed(0)
    B, Tg = 64, 7
    m = vt.attn_masks(ga, B, Tg, torch.device("cpu"))
    na, ne = 8 * Tg * Tg, Tg * 64
    assert m.shape == (B, na + ne + 3) and m.dtype == torch.float32
    keep = 1 / 0.9  # dropout 0.1 everywhere in the RTFS-Net configs (yaml:86)
    assert bool(((m == 0) | ((m - keep).abs() < 1e-6)).all())
    for seg in (m[:, :na], m[:, na:na + ne], m[:, na + ne:]):
        drop = float((seg == 0).float().mean())
        assert 0.03 < drop < 0.2, drop
    for mod in ga.modules():  # every probability 0: nothing to draw
        if isinstance(getattr(mod, "p", None), float):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    assert vt.attn_masks(ga, B, Tg, torch.device("cpu")) is None


def test_kink_helpers_on_the_oracle():
    _, sd, cfg = make_model(2)
    mix, _, emb = synth.synth_inputs(1, 32000, 50)
    assert _video_margin(sd, cfg, emb, False) < 2e-5  # the default full-length input sits 3e-6 rms from a video-branch kink ...
    e2 = stable_emb(sd, cfg, emb, False)
    assert _video_margin(sd, cfg, e2, False) >= 2e-5 and float((e2 - emb).abs().max()) < 1e-2  # ... a 1e-3 nudge moves it away
    mix, _, emb = synth.synth_inputs(2, 4096, 6)
    assert stable_emb(sd, cfg, emb, True) is emb  # nothing near a kink: the input itself
    sd2 = smooth_regime(sd, cfg, mix, emb, True)
    slopes = [float(v) for k, v in sd2.items() if k.endswith("weight") and v.numel() == 1]
    assert len(slopes) == 18 and all(0.969 < s < 1.001 for s in slopes)  # 2 + 12 + 1 audio-branch, 2 video-branch, 1 mask PReLU
    from oracle import avnet_ref

    seen = []
    avnet_ref.ACT_PROBE = lambda x, kind: seen.append(float(x.min())) if kind == "ReLU" else None
    try:
        with torch.no_grad():
            avnet_ref.avnet_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd2.items()}, cfg, mix.double(), emb.double(), training=True)
    finally:
        avnet_ref.ACT_PROBE = None
    assert len(seen) == 4 and min(seen) > 0  # every ReLU input positive on the smooth-regime weights


def test_bench_plain_launch_reexecutes_under_torchrun():
    """`python bench.py --gpus 2` without WORLD_SIZE: the script becomes `torch.distributed.run --nproc-per-node 2 bench.py ...`; without a GPU both
    ranks then stop at the device check (the product path has no CPU fallback) - what matters here is that they were started."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        return  # (on a GPU box tests/test_bench_contract.py exercises the real thing)
    assert r.returncode != 0
    assert "bench.py needs an MI355X" in (r.stdout + r.stderr)  # (the elastic agent may stop the second rank as soon as the first has failed)


def test_bench_traffic_is_null_when_the_kernel_source_changed(tmp_path, monkeypatch):
    """`roofline.traffic` comes from committed PMC passes (profiles/pmc_traffic.json); the file carries the hashes of the kernel sources it
    was measured with and bench.py reports null - with the reason - once one of them differs"""
    import hashlib
    import json
    import os
    import sys
    from types import SimpleNamespace

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    a = SimpleNamespace(layers=6, batch=32, seconds=2.0, dtype="f32", mode="infer")
    src = tmp_path / "rtfs_net_amd" / "csrc"
    src.mkdir(parents=True)
    (src / "dualpath.hip").write_text("kernel v1")
    (tmp_path / "profiles").mkdir()
    table = {"void rtfs::unfold_ffa_kernel<3, 0>(...)": {"bytes_per_launch": 300e6, "launches": 2},
             "void rtfs::unfold_ffa_kernel<4, 0>(...)": {"bytes_per_launch": 320e6, "launches": 2},
             "_source": {"rtfs_net_amd/csrc/dualpath.hip": hashlib.sha256(b"kernel v1").hexdigest()}}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(table))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got, why = bench.pmc_traffic("unfold_ffa_kernel<3, 0>|unfold_ffa_kernel<4, 0>", a)
    assert got == 310e6 and "committed PMC passes" in why
    (src / "dualpath.hip").write_text("kernel v2")
    got, why = bench.pmc_traffic("unfold_ffa_kernel<3, 0>|unfold_ffa_kernel<4, 0>", a)
    assert got is None and why.startswith("stale: rtfs_net_amd/csrc/dualpath.hip")
    del table["_source"]
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(table))
    assert bench.pmc_traffic("unfold_ffa_kernel<3, 0>", a) == (None, "profiles/pmc_traffic.json carries no source hash (measured before round 6): re-run tools/pmc_hbm.sh")
    a.batch = 16
    assert bench.pmc_traffic("unfold_ffa_kernel<3, 0>", a)[0] is None


def test_one_environment_variable_for_the_ab_switches(monkeypatch):
    """round 6: RTFS_DISABLE=<list> (was thirteen RTFS_NO_*_FUSION variables) and RTFS_VARIANTS=<family>:<n>; unknown names raise instead of being ignored"""
    import pytest

    from util import make_model

    monkeypatch.setenv("RTFS_DISABLE", "dwadj, wgside,vp_hip")
    monkeypatch.setenv("RTFS_VARIANTS", "resid:2")
    m, _, _ = make_model(2)
    assert not m._hip.fuse["dwadj"] and not m._hip.fuse["wgside"] and m._hip.fuse["trio"] and m._hip.vp_glue and m._hip.vp_side_stream
    assert m._hip.variants == {"resid": 2, "unfold": 0}
    monkeypatch.setenv("RTFS_DISABLE", "no_such_switch")
    with pytest.raises(ValueError, match="unknown switch"):
        make_model(2)
    monkeypatch.delenv("RTFS_DISABLE")
    monkeypatch.setenv("RTFS_VARIANTS", "gemm:1")
    with pytest.raises(ValueError, match="unknown kernel family"):
        make_model(2)
