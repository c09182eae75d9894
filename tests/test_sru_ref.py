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
