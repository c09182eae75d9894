"""Frozen lip encoder: oracle vs the reference's golden vectors on CPU; HIP kernels vs both on the GPU."""
import numpy as np
import pytest
import torch

from oracle import synth
from oracle.lip_ref import frcnn_forward, lip_inputs
from rtfs_net_amd.models import videomodels
from tests.util import load_npz, rel

GOLD = load_npz("lip.npz")
CASES = ["a", "b", "c"]  # 2x5 frames of 88x88, 1x3 of 96x96, 1x2 of 45x51 (odd sizes)
TOL = 1e-3  # relative L2 on the embeddings (BASELINE.json: 1e-3 on the path's outputs); fp32 everywhere, observed ~1e-6


def _model(relu_type="prelu"):
    m = videomodels.FRCNNVideoModel(relu_type=relu_type, print_macs=False)
    m.eval()
    sd = synth.synth_state_dict(m.state_dict(), salt=3)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_golden(case):
    _, sd = _model()
    B, T, H, W = (int(v) for v in GOLD[f"{case}_shape"])
    taps = {}
    y = frcnn_forward(sd, lip_inputs(B, T, H, W), taps)
    assert rel(y, torch.from_numpy(GOLD[f"{case}_out"])) < 2e-5
    for k, v in taps.items():
        assert rel(v[:, ::7, ::3, ::3], torch.from_numpy(GOLD[f"{case}_{k}"])) < 2e-5, k


def test_state_dict_keys_are_the_references():
    m, sd = _model()
    assert sorted(f"{k}:{tuple(v.shape)}" for k, v in sd.items()) == list(GOLD["keys"])
    assert videomodels.get("frcnnvideomodel") is videomodels.FRCNNVideoModel
    with pytest.raises(ValueError):
        videomodels.get("nope")
    with pytest.raises(ValueError):
        videomodels.FRCNNVideoModel(backbone_type="shufflenet", print_macs=False)


def test_protocol_frozen_batchnorm_and_no_cpu_fallback(capsys):
    m, _ = _model()
    m.train()
    assert all(not b.training for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm))
    with pytest.raises(RuntimeError):  # the product path never computes on the CPU
        with torch.no_grad():
            m(lip_inputs(1, 2))
    m.get_MACs()
    assert abs(m.macs - 15807.9) < 0.1 and "Pretrained Video Backbone" in capsys.readouterr().out
    ck = {"model_state_dict": {**{k: v + 1 for k, v in m.state_dict().items() if v.is_floating_point()}, "tcn.x": torch.zeros(1)}}
    before = m.state_dict()["trunk.layer1.0.conv1.weight"].clone()
    videomodels.update_frcnn_parameter(m, ck["model_state_dict"])
    assert torch.equal(m.state_dict()["trunk.layer1.0.conv1.weight"], before + 1)
    assert all(not p.requires_grad for p in m.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_matches_golden_and_oracle(case):
    m, sd = _model()
    m = m.cuda()
    B, T, H, W = (int(v) for v in GOLD[f"{case}_shape"])
    x = lip_inputs(B, T, H, W)
    with torch.no_grad():
        y = m(x.cuda())
    assert y.shape == (B, 512, T)
    sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
    e_gold, e_or = rel(y, torch.from_numpy(GOLD[f"{case}_out"])), rel(y, frcnn_forward(sd64, x.double()))
    print(f"lip {case}: vs reference golden {e_gold:.2e}, vs float64 oracle {e_or:.2e}")
    assert e_gold < TOL and e_or < TOL


@pytest.mark.gpu
def test_hip_relu_variant_and_weight_refresh():
    m, sd = _model("relu")
    m = m.cuda()
    x = lip_inputs(1, 4)
    with torch.no_grad():
        y = m(x.cuda())
        assert rel(y, frcnn_forward(sd, x)) < TOL
        m.trunk.layer3[1].bn2.running_var.mul_(1.7)  # a changed statistic must reach the folded weights
        y2 = m(x.cuda())
    sd2 = {k: v.cpu() for k, v in m.state_dict().items()}
    assert rel(y2, frcnn_forward(sd2, x)) < TOL and rel(y2, y) > 1e-3


@pytest.mark.gpu
def test_hip_refuses_autograd():
    m, _ = _model()
    m = m.cuda()
    with pytest.raises(RuntimeError):
        m(lip_inputs(1, 2).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(3, 9, 7, 64, 64, 3, 1), (2, 11, 11, 64, 128, 3, 2), (2, 11, 11, 64, 128, 1, 2), (5, 6, 6, 256, 256, 3, 1),
                                 (7, 3, 3, 512, 512, 3, 1), (1, 22, 22, 128, 64, 1, 1)])
def test_conv_entry_point(cfg):
    """rtfs_conv_nhwc_fwd against F.conv2d (float64 on the CPU): strides, kernel sizes, bias / residual / PReLU epilogue, ragged tiles."""
    from rtfs_net_amd import lib

    N, H, W, Cin, Cout, ks, stride = cfg
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + ks)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    bias, slope = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=ks // 2)
    res = torch.randn(ref.shape, generator=g)
    Ho, Wo = ref.shape[2:]
    xd, wk = x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().cuda()
    for use_b, use_s, use_r in ((False, False, False), (True, True, True), (True, False, True)):
        out = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda")
        lib.call("rtfs_conv_nhwc_fwd", xd, wk, bias.cuda() if use_b else None, slope.cuda() if use_s else None,
                 res.permute(0, 2, 3, 1).contiguous().cuda() if use_r else None, out, N, H, W, Cin, Cout, ks, stride)
        want = ref + (bias.double().view(1, -1, 1, 1) if use_b else 0) + (res.double() if use_r else 0)
        if use_s:
            want = torch.where(want >= 0, want, slope.double().view(1, -1, 1, 1) * want)
        assert rel(out.permute(0, 3, 1, 2), want) < 1e-5
    with pytest.raises(RuntimeError):
        lib.call("rtfs_conv_nhwc_fwd", xd, wk, None, None, None, out, N, H, W, Cin, Cout, 5, stride)
