"""GPU diagnostic: rtfs_gln_bwd_reduce / _apply (C = 256, ReLU after the norm: the audio bottleneck's pre-norm) against float64 torch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def main(B=1, rows=32379, C=256, act=2):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, rows, C, generator=g) * (0.5 + torch.rand(1, 1, C, generator=g)) + torch.randn(1, 1, C, generator=g)
    dy = torch.randn(B, rows, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = torch.nn.functional.group_norm(x64.permute(0, 2, 1), 1, g64, b64, 1e-5)
    y = torch.relu(y) if act == 2 else y
    (y * dy.double().permute(0, 2, 1)).sum().backward()
    xd, dyd = x.cuda().contiguous(), dy.cuda().contiguous()
    stats = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
    stats[:, 0] = xd.double().sum((1, 2))
    stats[:, 1] = (xd.double() ** 2).sum((1, 2))
    red = torch.zeros(B, lib.STAT_STRIDE, dtype=torch.float64, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    lib.call("rtfs_gln_bwd_reduce", dyd, xd, stats, gamma.cuda(), beta.cuda(), act, 0.0, red, dg, db, None, B, rows, C)
    dx = torch.empty_like(xd)
    lib.call("rtfs_gln_bwd_apply", dyd, xd, stats, gamma.cuda(), beta.cuda(), act, 0.0, red, dx, 0, B, rows, C)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())  # noqa: E731
    print(f"B={B} rows={rows} C={C} act={act}: dgamma rel {rel(dg, g64.grad):.3e}  dbeta rel {rel(db, b64.grad):.3e}  dx rel {rel(dx, x64.grad):.3e}")


if __name__ == "__main__":
    main()
    main(act=0)
    main(B=2, rows=4257, C=256)
    main(B=1, rows=32379, C=64, act=0)
