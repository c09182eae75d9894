"""Utterance-level data parallelism helpers (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

The separation path has no cross-utterance dependency, so the N>1 forward path shards the global
batch contiguously and exchanges NOTHING; the only collectives are the timing barrier and the reductions below.
"""
from __future__ import annotations

import torch


def shard_bounds(global_batch: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device, dist=None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device, dist=None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_outputs(local_out: torch.Tensor, global_batch: int, dist=None) -> torch.Tensor:
    """All-gather the separated waveforms of every shard back into global-batch order (evaluation helper)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_out
    world = dist.get_world_size()
    sizes = [shard_bounds(global_batch, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
    pad[: local_out.shape[0]] = local_out
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)
