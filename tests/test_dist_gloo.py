"""CPU, world_size 2, gloo: the N>1 path's sharding / reductions (no GPU needed; RCCL replaces gloo on the box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rtfs_net_amd.dist_util import gather_outputs, max_over_ranks, shard_bounds, sum_over_ranks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, gb, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(gb, rank, world)
    # a fake "separation": waveform b is filled with its global utterance index
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 1, 8).contiguous()
    full = gather_outputs(local, gb, dist)
    mx = max_over_ranks(1.0 + rank, "cpu", dist)
    sm = sum_over_ranks(float(hi - lo), "cpu", dist)
    q.put((rank, full[:, 0, 0].tolist(), mx, sm))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    for gb in (1, 7, 32, 255):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, gb = _free_port(), 7
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, full, mx, sm in res:
        assert full == [float(i) for i in range(gb)]
        assert mx == 2.0 and sm == float(gb)


# ---- SyncBatchNorm math of the CAF key/value embeddings across ranks (train.py:145 sync_batchnorm=True) ----------------------
def _bn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtfs_net_amd.models.hip_train import caf_bn_adjoint, caf_bn_batch_stats

    g = torch.Generator().manual_seed(5)
    C, ns = 6, (40, 25)  # ragged shards
    xs = [torch.randn(n, C, generator=g, dtype=torch.float64) * 2 + 0.5 for n in ns]
    dys = [torch.randn(n, C, generator=g, dtype=torch.float64) for n in ns]
    dw, gm, be = (torch.randn(C, generator=g, dtype=torch.float64) for _ in range(3))
    eps = 1e-5
    # this rank's share, as the HIP path sees it: per-channel sums only
    x, dy = xs[rank], dys[rank]
    sums = torch.stack([x.sum(0), (x * x).sum(0)])
    mean, var, n, lsum, lcov = caf_bn_batch_stats(sums, x.shape[0], True)
    inv = torch.rsqrt(dw.float() ** 2 * var + eps)
    c1, c2, c3, dgamma, dbeta, gdw = caf_bn_adjoint(dy.sum(0).float(), (dy * x).sum(0).float(), lsum, lcov, n, mean, var, dw.float(), gm.float(), inv, True)
    dx = dy.float() * c1 + c2 + c3 * x.float()
    pg = torch.stack([gdw, dgamma, dbeta])
    dist.all_reduce(pg)  # DDP would average; the sum must equal the single-process gradient
    # single-process truth over the union of both shards
    X = torch.cat(xs).requires_grad_(True)
    p = [t.clone().requires_grad_(True) for t in (dw, gm, be)]
    u = X * p[0]
    y = (u - u.mean(0)) / torch.sqrt(u.var(0, unbiased=False) + eps) * p[1] + p[2]
    (y * torch.cat(dys)).sum().backward()
    lo = sum(ns[:rank])
    q.put((rank, float((dx.double() - X.grad[lo:lo + ns[rank]]).abs().max()), float((pg.double() - torch.stack([t.grad for t in p])).abs().max()), n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sync_batchnorm_adjoint():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, edx, epg, n in res:
        assert n == 65.0
        assert edx < 1e-4 and epg < 1e-3, (rank, edx, epg)


# ---- SyncBatchNorm utterance count of the VP block's HIP training step (unequal last batch) ------------------------------
def _count_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtfs_net_amd.models.vp_train import BatchCountProbe

    outcome = []
    # three steps: equal batches, equal again (a per-rank cache of verified sizes would now skip the collective on rank 0 only), ragged last batch.
    # Every rank issues exactly one count all-reduce per step, followed by a statistics-shaped float64 all-reduce: the pairing must stay intact.
    for step, per_rank in enumerate([(8, 8), (8, 8), (8, 4)]):
        probe = BatchCountProbe(per_rank[rank], torch.device("cpu"))
        stats = torch.full((2, 64), float(rank + 1), dtype=torch.float64)
        dist.all_reduce(stats)
        assert float(stats[0, 0]) == 3.0
        try:
            probe.verify()
            outcome.append("ok")
        except ValueError as e:
            outcome.append("ValueError" if "equal per-rank batch sizes" in str(e) else repr(e))
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_unequal_last_batch_is_refused_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_count_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outcome in res:
        assert outcome == ["ok", "ok", "ValueError"], (rank, outcome)
