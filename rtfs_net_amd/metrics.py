"""Evaluation metrics of the reference that live next to the path: SI-SDR / SDR and their improvements over the
mixture, computed as `ALLMetricsTracker.__call__` does (src/metrics/allwrapper.py:35-55: PIT-wrapped `pairwise_neg_sisdr` /
`pairwise_neg_snr` of the estimate minus the same of the mixture repeated per source) on the HIP loss head (`rtfs_net_amd.losses`).
PESQ / STOI (third-party CPU packages, allwrapper.py:57-70) are outside the path and not built."""
from __future__ import annotations

import torch

from .losses import PITLossWrapper, pairwise_neg_sisdr, pairwise_neg_snr

_pit_sisnr = PITLossWrapper(pairwise_neg_sisdr, pit_from="pw_mtx")
_pit_snr = PITLossWrapper(pairwise_neg_snr, pit_from="pw_mtx")


def separation_metrics(mix: torch.Tensor, clean: torch.Tensor, estimate: torch.Tensor) -> dict:
    """mix [T], clean [n_src, T], estimate [n_src, T] on the GPU -> {"si-snr", "si-snr_i", "sdr", "sdr_i"} in dB (higher is better;
    the reference logs the negated losses, allwrapper.py:72-80)."""
    with torch.no_grad():
        est, cl = estimate.unsqueeze(0).float().contiguous(), clean.unsqueeze(0).float().contiguous()
        mx = torch.stack([mix] * clean.shape[0], dim=0).unsqueeze(0).float().contiguous()
        sisnr, sisnr_base = _pit_sisnr(est, cl), _pit_sisnr(mx, cl)
        sdr, sdr_base = _pit_snr(est, cl), _pit_snr(mx, cl)
    return {"si-snr": -float(sisnr), "si-snr_i": -float(sisnr - sisnr_base), "sdr": -float(sdr), "sdr_i": -float(sdr - sdr_base)}
