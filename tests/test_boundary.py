"""CPU: the drop-in boundary -- module API parity with src.models and the C-ABI library."""
import copy
import json
import os
import re

import pytest
import torch

from util import GOLDEN, ROOT, make_model, synth


def test_state_dict_keys_and_shapes_match_reference():
    model, _, _ = make_model(4)
    want = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(got) == list(want)  # same names, same order
    assert got == want
    assert sum(p.numel() for p in model.parameters()) == 740210  # 740,210 parameters


def test_window_buffers_are_non_persistent():
    model, _, _ = make_model(4)
    assert "encoder.window" not in model.state_dict() and "decoder.window" not in model.state_dict()
    assert model.encoder.window.shape == (256,)


def test_registry_like_src_models():
    import rtfs_net_amd.models as models

    assert models.get("avnet") is models.AVNet and models.get("RTFSNet") is models.AVNet
    with pytest.raises(ValueError):
        models.get("nope")
    with pytest.raises(ValueError):
        models.register_model(models.AVNet)

    class Custom(torch.nn.Module):
        pass

    models.register_model(Custom)
    assert models.get("custom") is Custom


def test_constructor_mutates_dicts_and_tolerates_unknown_keys():
    cfg = synth.rtfs_audionet(6)
    cfg["some_future_key"] = 1
    from rtfs_net_amd import AVNet

    m = AVNet(print_macs=False, **cfg)
    assert cfg["mask_generation_params"]["mask_generator_type"] == "MaskGenerator"  # tdavnet.py:54
    assert cfg["audio_bn_params"]["out_chan"] == 256  # tdavnet.py:56
    assert m.refinement_module.audio_net.repeats == 6


def test_unsupported_family_raises_value_error():
    from rtfs_net_amd import AVNet

    cfg = synth.rtfs_audionet(4)
    cfg["enc_dec_params"]["encoder_type"] = "ConvolutionalEncoder"
    with pytest.raises(ValueError):
        AVNet(print_macs=False, **cfg)
    with pytest.raises(ValueError):
        AVNet(print_macs=False, **copy.deepcopy(synth.TINY_AUDIONET))


def test_serialize_from_pretrain_roundtrip(tmp_path):
    from rtfs_net_amd import AVNet

    model, sd, cfg = make_model(4)
    conf = model.serialize()
    assert conf["model_name"] == "AVNet" and set(conf) == {"model_name", "state_dict", "model_args", "infos"}
    path = tmp_path / "best_model.pth"
    torch.save(conf, path)
    m2 = AVNet.from_pretrain(str(path), **copy.deepcopy(cfg))
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k])
    # Lightning checkpoints prefix the audio model's keys with "audio_model." (base_av_model.py:14-22)
    m3 = AVNet(print_macs=False, **copy.deepcopy(cfg))
    AVNet.load_state_dict_in(m3, {"audio_model." + k: v for k, v in sd.items()})
    assert torch.equal(m3.state_dict()["decoder.decoder.weight"], sd["decoder.decoder.weight"])


def test_macs_report_matches_published_table():
    model, _, _ = make_model(4)
    model.get_MACs()
    total = int(re.search(r"Total -+ MACs:\s+([\d,]+) M", model.macs_parms).group(1).replace(",", ""))
    assert abs(total - 21900) < 100  # docs/main_table.png: 21.9 G for RTFS-Net-4


def test_cpu_forward_fails_loudly():
    """the product path has no CPU fallback"""
    model, _, _ = make_model(4)
    mix, _, emb = synth.synth_inputs(1, 4000, 6)
    with pytest.raises(RuntimeError, match="HIP"):
        with torch.no_grad():
            model(mix, emb)


def test_library_exports_every_declared_symbol():
    """dlopen librtfs_hip.so and resolve every `int rtfs_*(` of include/rtfs_hip.h (no compute without a GPU)"""
    from rtfs_net_amd import lib

    header = open(os.path.join(ROOT, "include", "rtfs_hip.h")).read()
    declared = set(re.findall(r"^int (rtfs_\w+)\(", header, flags=re.M))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    handle = lib.load()
    for name in declared:
        assert getattr(handle, name) is not None


def test_vp_block_matches_oracle_on_cpu():
    """the only torch-executed piece of the product (video branch glue) against the oracle"""
    from oracle.avnet_ref import P, tdanet_block

    model, sd, cfg = make_model(4)
    _, _, emb = synth.synth_inputs(2, 16000, 25)
    with torch.no_grad():
        got = model.refinement_module.video_net.get_block(0)(emb)
        ref = tdanet_block(emb, P(sd, "refinement_module.video_net.blocks."), cfg["video_params"])
    assert float((got - ref).norm() / ref.norm()) < 1e-5


def test_ctypes_signatures_match_header():
    """argument count and kind (pointer / int / float / double / long long) of every binding equals the C prototype"""
    import ctypes

    from rtfs_net_amd import lib

    header = open(os.path.join(ROOT, "include", "rtfs_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = dict(re.findall(r"^int (rtfs_\w+)\(([^;]*?)\);", header, flags=re.M | re.S))
    assert set(protos) == set(lib.SIGNATURES)
    kind = {ctypes.c_void_p: "P", ctypes.c_int: "I", ctypes.c_float: "F", ctypes.c_longlong: "L", ctypes.c_double: "D"}
    for name, args in protos.items():
        want = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",")]:
            if a == "void":
                continue
            if "*" in a:
                want.append("P")
            elif a.startswith("long long"):
                want.append("L")
            elif a.startswith("float"):
                want.append("F")
            elif a.startswith("double"):
                want.append("D")
            elif a.startswith("int"):
                want.append("I")
            else:
                raise AssertionError(f"{name}: cannot classify '{a}'")
        got = [kind[t] for t in lib.SIGNATURES[name]]
        assert got == want, (name, "".join(got), "".join(want))


def test_models_subpackage_is_relocatable(tmp_path):
    """train.py:95 copies `src/models` into the experiment directory and test.py:33-36 imports it back as `<exp>.models`
    (a plain directory import, no parent package around it): a copy of rtfs_net_amd/models must import and build the model there."""
    import copy
    import importlib
    import shutil
    import sys

    from rtfs_net_amd import synthetic

    exp = tmp_path / "exp_relocated"
    shutil.copytree(os.path.join(ROOT, "rtfs_net_amd", "models"), exp / "models", ignore=shutil.ignore_patterns("__pycache__"))
    sys.path.append(str(tmp_path))
    try:
        mod = importlib.import_module("exp_relocated.models")
        vm = importlib.import_module("exp_relocated.models.videomodels")
        assert mod.AVNet.__module__.startswith("exp_relocated.models") and hasattr(vm, "FRCNNVideoModel")
        model = mod.AVNet(print_macs=False, **copy.deepcopy(synthetic.rtfs_audionet(2)))
        assert sum(p.numel() for p in model.parameters()) == 740210
    finally:
        sys.path.remove(str(tmp_path))
        for k in [k for k in sys.modules if k.startswith("exp_relocated")]:
            del sys.modules[k]
