"""CPU: internal consistency of the SRU restatement (oracle/sru_ref.py; 'parity unpinned' -- the sru package is absent)."""
import torch

from oracle.sru_ref import SRU, sru_cell_forward


def test_shapes_and_matrix_count():
    rnn = SRU(512, 32, num_layers=4, bidirectional=True)
    assert [tuple(c.weight.shape) for c in rnn.rnn_lst] == [(512, 256), (64, 192), (64, 192), (64, 192)]
    h, c = rnn(torch.randn(7, 3, 512))
    assert h.shape == (7, 3, 64) and c.shape == (4, 3, 64)


def test_backward_direction_is_time_reversed_forward_direction():
    """running direction 1 on x equals running direction 0 on flip(x) with the direction-1 parameters"""
    g = torch.Generator().manual_seed(0)
    L, B, d = 11, 2, 4
    x = torch.randn(L, B, 2 * d, generator=g)
    W = torch.randn(2 * d, 2 * d * 3, generator=g)
    wc, b = torch.randn(4 * d, generator=g), torch.randn(4 * d, generator=g)
    h, _ = sru_cell_forward(x, W, wc, b, torch.ones(1), d, True)
    # swap the two directions' parameters, feed the flipped sequence
    Wv = W.view(2, d, 2, d, 3).flip(0).flip(2).reshape(2 * d, -1)  # rows follow the swapped input halves
    wcv, bv = wc.view(2, 2, d).flip(1).reshape(-1), b.view(2, 2, d).flip(1).reshape(-1)
    xs = x.view(L, B, 2, d).flip(2).reshape(L, B, 2 * d)  # skip term x' follows its direction's slice
    h2, _ = sru_cell_forward(xs.flip(0), Wv, wcv, bv, torch.ones(1), d, True)
    assert torch.allclose(h.view(L, B, 2, d)[:, :, 1], h2.flip(0).view(L, B, 2, d)[:, :, 0], atol=1e-6)


def test_single_step_closed_form():
    d = 3
    x = torch.randn(1, 1, 2 * d)
    W = torch.randn(2 * d, 2 * d * 3)
    wc, b = torch.randn(4 * d), torch.randn(4 * d)
    h, c = sru_cell_forward(x, W, wc, b, torch.ones(1), d, True)
    U = (x[0] @ W).view(1, 2, d, 3)
    f = torch.sigmoid(U[..., 1] + b.view(2, 2, d)[0])
    r = torch.sigmoid(U[..., 2] + b.view(2, 2, d)[1])
    c1 = U[..., 0] * (1 - f)
    h1 = x.view(1, 2, d) * (1 - r) + c1 * r
    assert torch.allclose(h.view(1, 2, d), h1, atol=1e-6) and torch.allclose(c.view(1, 2, d), c1, atol=1e-6)


def test_c_restatement_of_the_recurrence_matches_the_python_loop():
    """oracle/csrc/sru_scan.c (what the oracle runs without autograd: forward parity checks, bench.py's CPU baseline) against the Python loop, k = 3 and k = 4,
    float32 and float64, both directions, scale_x != 1"""
    import pytest

    from oracle.sru_ref import _c_scan

    if _c_scan() is None:
        pytest.skip("no gcc: the oracle keeps its Python loop")
    g = torch.Generator().manual_seed(3)
    for d_in, dtype, tol in ((64, torch.float32, 2e-6), (512, torch.float32, 2e-6), (64, torch.float64, 1e-14), (512, torch.float64, 1e-14)):
        L, B, d = 23, 5, 32
        k = 3 if d_in == 2 * d else 4
        x = torch.randn(L, B, d_in, generator=g).to(dtype)
        W = (torch.randn(d_in, 2 * d * k, generator=g) * 0.1).to(dtype)
        wc, b = torch.randn(4 * d, generator=g).to(dtype), torch.randn(4 * d, generator=g).to(dtype)
        sx = torch.tensor([0.7], dtype=dtype)
        with torch.no_grad():
            h_c, c_c = sru_cell_forward(x, W, wc, b, sx, d, True)
            h_p, c_p = sru_cell_forward(x, W, wc, b, sx, d, True, use_c=False)
        assert float((h_c - h_p).abs().max()) < tol and float((c_c - c_p).abs().max()) < tol, (d_in, dtype)
    # under autograd the Python loop runs (gradients flow)
    x = torch.randn(5, 2, 64, generator=g).requires_grad_(True)
    W = torch.randn(64, 192, generator=g) * 0.1
    h, _ = sru_cell_forward(x, W, torch.randn(128, generator=g), torch.randn(128, generator=g), torch.ones(1), 32, True)
    h.sum().backward()
    assert x.grad is not None and float(x.grad.abs().max()) > 0


def _real_sru():
    """the third-party package if the environment has it (never the import stub of oracle/stubs); None otherwise"""
    import importlib
    import os
    import sys

    from oracle.ref_import import STUBS

    saved = list(sys.path)
    try:
        sys.path[:] = [p for p in sys.path if os.path.realpath(p) != os.path.realpath(STUBS)]
        cached = sys.modules.get("sru")
        if cached is not None and os.path.realpath(getattr(cached, "__file__", "") or "").startswith(os.path.realpath(STUBS)):
            del sys.modules["sru"]
        try:
            return importlib.import_module("sru")
        except ImportError:
            return None
    finally:
        sys.path[:] = saved


def test_restatement_matches_the_package():
    """THE pin of SURVEY.md §8 a7 (rnn_layers.py:6,100-105,150; setup/requirements.yaml:18,33): oracle/sru_ref.py against the real `sru` package - forward and
    float64 autograd, k = 4 (input 512) and k = 3 (input 64) layers, bidirectional, default options as the reference constructs it and non-default
    highway_bias / rescale / scale_x.  Skipped where the package is absent (this container: no distribution, no network); `pip install sru==2.6.0` closes it."""
    import pytest

    pkg = _real_sru()
    if pkg is None:
        pytest.skip("the `sru` package is not installed: SRU parity stays 'unpinned' (oracle/ref_import.py says how to close it)")
    g = torch.Generator().manual_seed(11)
    for kwargs in ({}, {"highway_bias": -1.5}, {"rescale": True}, {"rescale": True, "highway_bias": 0.7}):
        for d_in, layers in ((512, 4), (64, 2)):
            real = pkg.SRU(input_size=d_in, hidden_size=32, num_layers=layers, bidirectional=True, **kwargs).double()
            mine = SRU(d_in, 32, num_layers=layers, bidirectional=True, **kwargs).double()
            sd = {k: v.clone() for k, v in real.state_dict().items()}
            assert set(sd) == set(mine.state_dict()), (set(sd) ^ set(mine.state_dict()))  # same key names: checkpoints interchange
            for k in sd:  # random recurrent vectors / biases (the package initialises the bias to the constant highway_bias)
                if k.endswith(("weight_c", "bias")):
                    sd[k] = sd[k] + 0.5 * torch.randn(sd[k].shape, generator=g, dtype=torch.float64)
            if not kwargs:
                sd = {k: (torch.full_like(v, 0.8) if k.endswith("scale_x") else v) for k, v in sd.items()}  # a checkpoint with scale_x != 1
            real.load_state_dict(sd), mine.load_state_dict(sd)
            x = torch.randn(13, 3, d_in, generator=g, dtype=torch.float64)
            xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            (ha, ca), (hb, cb) = real(xa), mine(xb)
            assert float((ha - hb).abs().max()) < 1e-12 and float((ca - cb).abs().max()) < 1e-12, (kwargs, d_in)
            w = torch.randn(ha.shape, generator=g, dtype=torch.float64)
            (ha * w).sum().backward(), (hb * w).sum().backward()
            assert float((xa.grad - xb.grad).abs().max()) < 1e-11
            for (n, pa), (_, pb) in zip(real.named_parameters(), mine.named_parameters()):
                assert float((pa.grad - pb.grad).abs().max()) < 1e-10 * max(1.0, float(pa.grad.abs().max())), (kwargs, n)
            with torch.no_grad():  # fp32, what the fixtures run
                h32a, h32b = real.float()(x.float())[0], mine.float()(x.float())[0]
            assert float((h32a - h32b).abs().max()) < 1e-5


def test_generators_use_an_installed_sru_package_and_record_it(tmp_path):
    """with an `sru` package in the environment the fixture generators import IT (the stub directory is last on sys.path, it used to be
    first) and record `sru_source = "package sru <version>"`; without one they fall back to the restatement and say so.  The package here is a fake that
    marks itself; every committed SRU fixture carries the key."""
    import os
    import subprocess
    import sys

    import numpy as np

    from oracle.ref_import import RESTATEMENT
    from oracle.regenerate_all import sru_fixture_files

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "site" / "sru"
    fake.mkdir(parents=True)
    (fake / "__init__.py").write_text("from oracle.sru_ref import SRU, SRUCell\n__version__ = '9.9.9+fake'\nMARK = 'environment package'\n")
    code = ("from oracle.ref_import import prepare_path, sru_source\nprepare_path()\nimport sru\n"
            "print(getattr(sru, 'MARK', 'stub'), '|', sru_source())\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path / "site"), root]))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.strip()
    assert out == "environment package | package sru 9.9.9+fake", out
    env = dict(os.environ, PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.strip()
    if _real_sru() is None:
        assert out == f"stub | {RESTATEMENT}", out
    files = sru_fixture_files()
    assert len(files) >= 22
    for path in files:
        with np.load(path) as z:
            assert "sru_source" in z.files, path
            src = str(z["sru_source"])
        assert src == RESTATEMENT or src.startswith("package sru "), (path, src)
