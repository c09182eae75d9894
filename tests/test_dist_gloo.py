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
