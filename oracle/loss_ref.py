"""ORACLE (test infrastructure - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this).

CPU restatement of the loss / PIT head that follows the separation path:
  pairwise_neg_sdr   src/losses/matrix.py:13-53   (PairwiseNegSDR.forward; sdr_type snr | sisdr | sdsdr)
  pit_pw_mtx         src/losses/pit_wrapper.py:26-51,82-107 (PITLossWrapper, pit_from="pw_mtx", factorial search)
Pinned against the reference itself by oracle/gen_golden_loss.py -> tests/golden/loss.npz.
"""
from itertools import permutations

import torch

EPS = 1e-8


def pairwise_neg_sdr(ests, targets, sdr_type="snr", zero_mean=True, take_log=True):
    """ests, targets [B, n_src, T] -> [B, n_src(est), n_src(target)]   (matrix.py:21-53)"""
    if targets.size() != ests.size() or targets.ndim != 3:
        raise TypeError(f"Inputs must be of shape [batch, n_src, time], got {ests.size()} and {targets.size()} instead")
    if zero_mean:  # matrix.py:26-30
        targets = targets - targets.mean(2, keepdim=True)
        ests = ests - ests.mean(2, keepdim=True)
    s_t, s_e = targets.unsqueeze(1), ests.unsqueeze(2)  # matrix.py:32-33
    if sdr_type in ("sisdr", "sdsdr"):  # matrix.py:34-40
        dot = (s_e * s_t).sum(3, keepdim=True)
        energy = (s_t ** 2).sum(3, keepdim=True) + EPS
        proj = dot * s_t / energy
    else:  # matrix.py:41-43
        proj = s_t.repeat(1, s_t.shape[2], 1, 1)
    noise = s_e - s_t if sdr_type in ("sdsdr", "snr") else s_e - proj  # matrix.py:44-47
    sdr = (proj ** 2).sum(3) / ((noise ** 2).sum(3) + EPS)  # matrix.py:49
    if take_log:
        sdr = 10 * torch.log10(sdr + EPS)  # matrix.py:51
    return -sdr


def pit_pw_mtx(pw_loss):
    """pw_loss [B, n_est, n_tgt] -> (mean over batch of the best permutation's mean loss, best permutation [B, n_src])
    (pit_wrapper.py:42-47,82-107: transpose, one-hot einsum over all permutations, / n_src, min)"""
    n = pw_loss.shape[-1]
    pwl = pw_loss.transpose(-1, -2)
    perms = torch.tensor(list(permutations(range(n))), dtype=torch.long)
    loss_set = torch.stack([sum(pwl[:, i, p[i]] for i in range(n)) / n for p in perms], 1)
    min_loss, idx = loss_set.min(1)
    return min_loss.mean(), perms[idx]
