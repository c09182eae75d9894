"""ORACLE (test infrastructure, not product code): CPU restatement of the Simple Recurrent Unit.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The reference calls a third-party package that is NOT vendored in /root/reference:

    from sru import SRU                                   (src/models/layers/rnn_layers.py:6)
    SRU(input_size=512, hidden_size=32, num_layers=4,
        bidirectional=True)                               (src/models/layers/rnn_layers.py:100-105)
    x = self.rnn(x)[0]                                    (src/models/layers/rnn_layers.py:150)

Pin: setup/requirements.yaml:33 (git+https://github.com/taolei87/sru.git, HEAD), alternative pin
`sru==2.6.0` at setup/requirements.yaml:18.  The package is absent here and cannot be installed
(no network), and the reference holds no test or golden vector at this call site, so

    *** PARITY UNPINNED for the SRU arithmetic ***

How to pin it, on any box with the package (`pip install sru==2.6.0`) - nothing has to be edited:
    python -m pytest tests/test_sru_ref.py::test_restatement_matches_the_package     (this file against the package: forward + float64 autograd)
    python -m oracle.regenerate_all                                                   (the generators import the installed package before the stub,
                                                                                      oracle/ref_import.py; fixtures then say sru_source = "package sru 2.6.0")

What follows restates the published algorithm of sru 2.6.0 (`SRUCell.forward` +
`elementwise_recurrence_naive`, "Simple Recurrent Units for Highly Parallelizable Recurrence",
Lei et al. 2018) with the constructor defaults the reference relies on:

    dropout=0, rnn_dropout=0, projection_size=0, use_tanh=False, layer_norm=False,
    has_skip_term=True, highway_bias=0.0, rescale=False, v1=False

Per layer (d = hidden_size, D = number of directions, k = 4 if input_size != D*d else 3):

    U = x @ weight                        weight: [input_size, D*d*k]; column ((dir*d + j)*k + m)
    wf, wr = weight_c.view(2, D, d)       recurrent ("peephole") vectors
    bf, br = bias.view(2, D, d)
    x' = U[..., 3]            if k == 4   (learned skip projection)
       = x * scale_x          if k == 3   (scale_x buffer is 1 when rescale=False)
    direction 0 scans t = 0..L-1, direction 1 scans t = L-1..0 over the SAME (unflipped) input:
        f_t = sigmoid(U1_t + bf + wf * c_{t-1})
        r_t = sigmoid(U2_t + br + wr * c_{t-1})
        c_t = U0_t + (c_{t-1} - U0_t) * f_t
        h_t = x'_t + (c_t - x'_t) * r_t
    c_{-1} = 0; output h: [L, B, D*d] = concat(dir0, dir1); c_last: [B, D*d]

`rescale` and `highway_bias` are constructor options here so that a checkpoint trained with a
different sru revision can still be matched.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


_C_SCAN = None  # ctypes handle of oracle/_build/libsru_scan.so (False: unavailable)


def _c_scan():
    """the C restatement of the recurrence (oracle/csrc/sru_scan.c), built on first use; None without gcc"""
    global _C_SCAN
    if _C_SCAN is None:
        try:
            import ctypes

            from .build_c import build

            path = build()
            lib = ctypes.CDLL(path) if path else None
            if lib is not None:
                for fn in (lib.sru_scan_f32, lib.sru_scan_f64):
                    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2 + [ctypes.c_int]
                    fn.restype = None
            _C_SCAN = lib or False
        except Exception:  # noqa: BLE001  (the Python loop below is the same arithmetic)
            _C_SCAN = False
    return _C_SCAN or None


def _scan_in_c(U, xp, weight_c, bias, L, B, D, d, k):
    """h [L,B,D,d], c_last [B,D,d] from the C loop; inference only (no autograd), CPU float32 / float64"""
    lib = _c_scan()
    fn = lib.sru_scan_f32 if U.dtype == torch.float32 else lib.sru_scan_f64
    U, wc, bs = U.contiguous(), weight_c.detach().to(U.dtype).contiguous(), bias.detach().to(U.dtype).contiguous()
    xpc = None if xp is None else xp.contiguous()
    h = torch.empty(L, B, D, d, dtype=U.dtype)
    c_last = torch.empty(B, D, d, dtype=U.dtype)
    fn(U.data_ptr(), 0 if xpc is None else xpc.data_ptr(), wc.data_ptr(), bs.data_ptr(), L, B, D, d, k, h.data_ptr(), c_last.data_ptr(), max(1, min(torch.get_num_threads(), B * D)))
    return h, c_last


def sru_cell_forward(x, weight, weight_c, bias, scale_x, hidden_size, bidirectional=True, c0=None, use_c=True):
    """One SRU layer, time-major. x: [L, B, d_in] -> (h [L, B, D*d], c_last [B, D*d]).  Without autograd (forward parity checks, bench.py's CPU baseline) the
    recurrence runs in the C restatement (oracle/csrc/sru_scan.c: same formulas, same order); under autograd, or without gcc, in the Python loop below."""
    L, B, d_in = x.shape
    D = 2 if bidirectional else 1
    d = hidden_size
    k = weight.shape[1] // (D * d)
    U = (x.reshape(L * B, d_in) @ weight).view(L, B, D, d, k)
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (x, weight, weight_c, bias))
    if (use_c and not needs_grad and c0 is None and x.device.type == "cpu" and U.dtype in (torch.float32, torch.float64) and d <= 256 and _c_scan() is not None):
        xp_c = (x.view(L, B, D, d) * scale_x).to(U.dtype) if k == 3 else None
        h, c_last = _scan_in_c(U.detach(), None if xp_c is None else xp_c.detach(), weight_c, bias, L, B, D, d, k)
        return h.view(L, B, D * d), c_last.reshape(B, D * d)
    wf, wr = weight_c.view(2, D, d)
    bf, br = bias.view(2, D, d)
    if k == 3:
        xp = x.view(L, B, D, d) * scale_x
    else:
        xp = U[..., 3]
    h = x.new_zeros(L, B, D, d)
    c_init = x.new_zeros(B, D, d) if c0 is None else c0.view(B, D, d)
    c_fin = []
    for di in range(D):
        order = range(L) if di == 0 else range(L - 1, -1, -1)
        c = c_init[:, di]
        u0 = U[:, :, di, :, 0]
        u1 = U[:, :, di, :, 1] + bf[di]
        u2 = U[:, :, di, :, 2] + br[di]
        for t in order:
            f = torch.sigmoid(u1[t] + c * wf[di])
            r = torch.sigmoid(u2[t] + c * wr[di])
            c = u0[t] + (c - u0[t]) * f
            h[t, :, di] = xp[t, :, di] + (c - xp[t, :, di]) * r
        c_fin.append(c)
    return h.view(L, B, D * d), torch.stack(c_fin, 1).reshape(B, D * d)


def sru_forward(x, layers, hidden_size, bidirectional=True):
    """Multi-layer SRU. `layers` = list of dicts {weight, weight_c, bias, scale_x}. Returns (h, c_stack)."""
    cs = []
    for p in layers:
        x, c = sru_cell_forward(x, p["weight"], p["weight_c"], p["bias"], p["scale_x"], hidden_size, bidirectional)
        cs.append(c)
    return x, torch.stack(cs, 0)


class SRUCell(nn.Module):
    """Parameter holder with the state-dict layout of sru.SRUCell (weight, weight_c, bias, scale_x)."""

    def __init__(self, input_size, hidden_size, bidirectional=True, highway_bias=0.0, rescale=False):
        super().__init__()
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        self.highway_bias, self.rescale = highway_bias, rescale
        D = 2 if bidirectional else 1
        self.output_size = hidden_size * D
        self.num_matrices = 3 if input_size == self.output_size else 4
        self.weight = nn.Parameter(torch.empty(input_size, self.output_size * self.num_matrices))
        self.weight_c = nn.Parameter(torch.empty(2 * self.output_size))
        self.bias = nn.Parameter(torch.empty(2 * self.output_size))
        self.register_buffer("scale_x", torch.ones(1))
        self.reset_parameters()

    def reset_parameters(self):
        d_in, d = self.input_size, self.output_size
        with torch.no_grad():
            self.weight.uniform_(-(3.0 / d_in) ** 0.5, (3.0 / d_in) ** 0.5)
            w = self.weight.view(d_in, d, self.num_matrices)
            w[:, :, 1].mul_(0.5**0.5)
            w[:, :, 2].mul_(0.5**0.5)
            self.weight_c.uniform_(-(3.0**0.5), 3.0**0.5).mul_(0.5**0.5)
            self.bias.zero_()
            self.bias[d:].add_(self.highway_bias)
            self.scale_x.fill_(1.0)
            if self.rescale:
                self.scale_x.fill_((1 + math.exp(self.highway_bias) * 2) ** 0.5)
                if self.num_matrices == 4:
                    w[:, :, 3].mul_(float(self.scale_x))

    def forward(self, x, c0=None):
        return sru_cell_forward(x, self.weight, self.weight_c, self.bias, self.scale_x, self.hidden_size, self.bidirectional, c0)


class SRU(nn.Module):
    """Stand-in with the constructor/return contract of sru.SRU used at rnn_layers.py:100-105,150."""

    def __init__(self, input_size, hidden_size, num_layers=2, bidirectional=False, highway_bias=0.0, rescale=False, **unused):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.bidirectional = input_size, hidden_size, num_layers, bidirectional
        out = hidden_size * (2 if bidirectional else 1)
        self.rnn_lst = nn.ModuleList(
            SRUCell(input_size if i == 0 else out, hidden_size, bidirectional, highway_bias, rescale) for i in range(num_layers)
        )

    def forward(self, x, c0=None):
        cs = []
        for cell in self.rnn_lst:
            x, c = cell(x)
            cs.append(c)
        return x, torch.stack(cs, 0)
