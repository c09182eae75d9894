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
