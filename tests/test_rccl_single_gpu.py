"""GPU: the N > 1 training path over RCCL itself, on the one GPU of the test box.

`init_process_group("nccl", world_size=1)` creates a real RCCL communicator; DistributedDataParallel then registers its bucket hooks
and runs its all-reduce on that communicator, SyncBatchNorm and the CAF batch-statistics reductions (models/hip_train.py) issue
their collectives through it - so the first time RCCL sees this code is not the driver's 8-GPU run.  With one rank every collective
is the identity, hence the gradients must equal the plain single-process step (train.py:135-146 of the reference)."""
import os

import pytest
import torch
import torch.distributed as dist

from util import make_model, rel, synth

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_ddp_syncbn_step_over_rccl_world_size_one():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        model, _, _ = make_model(2, "cuda")
        for mod in model.modules():  # dropout off: the two runs must see the same function
            if isinstance(getattr(mod, "p", None), float):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        model.train()
        B, L, Tv = 2, 4096, 12  # Tv >= 8: the VP block's HIP training kernels under SyncBatchNorm modules (world size 1: no collective)
        mix, _, emb = synth.synth_inputs(B, L, Tv)
        wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(11))
        mix, emb, wgt = mix.cuda(), emb.cuda(), wgt.cuda()
        model.zero_grad(set_to_none=True)
        ((model(mix, emb) * wgt).sum((1, 2)).mean()).backward()
        ref = {n: p.grad.clone() for n, p in model.named_parameters()}
        for mod in model.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.reset_running_stats()
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in net.modules())
        net = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], bucket_cap_mb=25)
        net.zero_grad(set_to_none=True)
        ((net(mix, emb) * wgt).sum((1, 2)).mean()).backward()
        # a collective of our own on the communicator (bench.py's max-over-ranks), then compare
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert float(t) == 1.5
        scale = max(float(g.norm()) for g in ref.values())
        for n, p in net.module.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
            err = float((p.grad - ref[n]).norm()) / (float(ref[n].norm()) + 1e-4 * scale)
            assert err < (2e-2 if p.numel() <= 12 else 5e-3), (n, err)
    finally:
        dist.destroy_process_group()
