"""Loss / PIT head of the training step, same API as the reference's `src.losses`:

    PairwiseNegSDR(sdr_type)          src/losses/matrix.py:13-53       -> [B, n_src, n_src] pair-wise negative SDR
    PITLossWrapper(loss, "pw_mtx")    src/losses/pit_wrapper.py:14-107 -> mean over the batch of the best permutation
    pairwise_neg_snr / pairwise_neg_sisdr / pairwise_neg_sdsdr         (matrix.py:141-143)

The pair-wise matrix and its adjoint are HIP kernels (csrc/loss.hip: the waveforms are read once for the six sums every formula
needs, once more for the gradient); the permutation search over the tiny [B, n, n] matrix is host logic in torch, as in the
reference (factorial search for n_src <= 3, Hungarian via scipy above).  CUDA tensors only: no CPU fallback.
"""
from __future__ import annotations

from itertools import permutations

import torch
import torch.nn as nn

from . import lib

_KINDS = {"snr": 0, "sisdr": 1, "sdsdr": 2}


class _PairwiseNegSDRFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ests, targets, kind, zero_mean, take_log):
        if not ests.is_cuda:
            raise RuntimeError("rtfs_net_amd.losses runs on an MI355X HIP device only (no CPU fallback)")
        e, t = ests.detach().float().contiguous(), targets.detach().float().contiguous()
        B, n, T = e.shape
        sums = torch.zeros(B * n * n * 6, dtype=torch.float64, device=e.device)
        pw = torch.empty(B, n, n, device=e.device)
        coef = torch.empty(B * n * n * 4, device=e.device)
        lib.call("rtfs_neg_sdr_sums", e, t, sums, B, n, T)
        lib.call("rtfs_neg_sdr_finish", sums, kind, 1 if zero_mean else 0, 1 if take_log else 0, pw, coef, B, n, T)
        ctx.save_for_backward(e, t, coef)
        return pw

    @staticmethod
    def backward(ctx, g):
        e, t, coef = ctx.saved_tensors
        B, n, T = e.shape
        dest = torch.empty_like(e)
        lib.call("rtfs_neg_sdr_grad", e, t, coef, g.float().contiguous(), dest, B, n, T)
        return dest, None, None, None, None


class PairwiseNegSDR(nn.Module):
    def __init__(self, sdr_type, zero_mean=True, take_log=True, EPS=1e-8):
        super().__init__()
        assert sdr_type in _KINDS
        if EPS != 1e-8:
            raise ValueError("the HIP loss head is built for the reference's EPS = 1e-8")
        self.sdr_type, self.zero_mean, self.take_log, self.EPS = sdr_type, zero_mean, take_log, EPS

    def forward(self, ests, targets):
        if targets.size() != ests.size() or targets.ndim != 3:
            raise TypeError(f"Inputs must be of shape [batch, n_src, time], got {ests.size()} and {targets.size()} instead")
        return _PairwiseNegSDRFn.apply(ests, targets, _KINDS[self.sdr_type], self.zero_mean, self.take_log)


_PERM_CACHE = {}


class PITLossWrapper(nn.Module):
    """pit_from="pw_mtx" (the only mode train.py / the metrics use: train.py:98-101, metrics/allwrapper.py:32-33)."""

    def __init__(self, loss_func, pit_from="pw_mtx", perm_reduce=None):
        super().__init__()
        if pit_from != "pw_mtx" or perm_reduce is not None:
            raise ValueError("only pit_from='pw_mtx' without perm_reduce is built (what the reference's training and evaluation use)")
        self.loss_func, self.pit_from = loss_func, pit_from

    def forward(self, ests, targets, return_ests=False, **kwargs):
        pw_loss = self.loss_func(ests, targets, **kwargs)
        assert pw_loss.ndim == 3 and pw_loss.shape[0] == targets.shape[0]
        min_loss, batch_indices = self.find_best_perm(pw_loss)
        mean_loss = torch.mean(min_loss)
        if not return_ests:
            return mean_loss
        return mean_loss, torch.stack([torch.index_select(s, 0, b) for s, b in zip(ests, batch_indices)])

    @staticmethod
    def find_best_perm(pair_wise_losses):
        n = pair_wise_losses.shape[-1]
        pwl = pair_wise_losses.transpose(-1, -2)
        if n <= 3:  # pit_wrapper.py:82-107
            # the permutation table and its one-hot form are constants of (n, device, dtype): built once.  Built per call (as the reference does), the
            # host-to-device copy of the table is a blocking transfer in the middle of the training step - the host then waits for the whole forward and the
            # device idles ~0.9 ms while the loss and the first backward launches are enqueued (round 5, tools/train_gaps.py)
            key = (n, pwl.device, pwl.dtype)
            cached = _PERM_CACHE.get(key)
            if cached is None:
                # built outside inference mode: a table first made under torch.inference_mode() (Lightning's sanity-check validation) would be an
                # inference tensor, which the first training step's einsum could not save for backward
                with torch.inference_mode(False), torch.no_grad():
                    perms = torch.tensor(list(permutations(range(n))), dtype=torch.long, device=pwl.device)
                    one_hot = torch.zeros((*perms.size(), n), dtype=pwl.dtype, device=pwl.device).scatter_(2, perms.unsqueeze(2), 1)
                cached = _PERM_CACHE[key] = (perms, one_hot)
            perms, one_hot = cached
            loss_set = torch.einsum("bij,pij->bp", pwl, one_hot) / n
            min_loss, idx = torch.min(loss_set, dim=1)
            return min_loss, perms[idx]
        from scipy import optimize  # pit_wrapper.py:109-117

        cpu = pwl.detach().cpu()
        batch_indices = torch.tensor([optimize.linear_sum_assignment(m)[1] for m in cpu]).to(pwl.device)
        return torch.gather(pwl, 2, batch_indices[..., None]).mean([-1, -2]), batch_indices


pairwise_neg_sisdr = PairwiseNegSDR("sisdr")
pairwise_neg_sdsdr = PairwiseNegSDR("sdsdr")
pairwise_neg_snr = PairwiseNegSDR("snr")
