"""GPU micro-benchmark of rtfs_resid_proj_fwd (TFAR tail + residual conv + next block's gateway / projection) at the bench shape; with a
-DRESID_TIMING build (tools/build_variant.sh residtime -DRESID_TIMING; RTFS_HIP_LIB=exp/residtime/librtfs_hip.so) also the per-wave s_memtime
segment sums of resid_ws_kernel:  M waves: conv, proj, -, -, barrier wait, loop head;  X waves: epilogue, load_sv issue, xform_e, load_e issue,
barrier wait, loop head.

    python tools/resid_bench.py [variant] [B]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtfs_net_amd import lib  # noqa: E402


def main(variant=0, B=32, T=251):
    F, F2, T2 = 129, 64, (T - 2) // 2 + 1
    TF = T * F
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    cl, d0 = rnd(B, TF, 64), rnd(B, TF, 64)
    cg, cgate = rnd(B, T2 * F2, 64), rnd(B, T2 * F2, 64)
    st = lambda n: torch.stack([torch.tensor([0.0, float(n)] + [0.0] * 14, dtype=torch.float64)] * B).cuda()  # noqa: E731
    sf, sl = st(TF * 64), st(T2 * F2 * 64)
    g64, b64 = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    Wt, bias = rnd(256, 64) * 0.1, rnd(256) * 0.1
    s_in, a0 = rnd(B, TF, 256), rnd(B, TF, 256)
    gw, gb = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    Wp, pbias = rnd(64, 256) * 0.06, rnd(64) * 0.1
    out, py = torch.empty(B, TF, 256, device="cuda"), torch.empty(B, TF, 64, device="cuda")
    pst = torch.zeros(B, 16, dtype=torch.float64, device="cuda")
    args = (cl, sf, g64, b64, d0, sf, g64, b64, cg, sl, g64, b64, cgate, sl, g64, b64, Wt, bias, s_in, gw, gb, 0.25, a0, out, Wp, pbias, py, pst, B, T, T2, variant)
    for _ in range(3):
        lib.call("rtfs_resid_proj_fwd", *args)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record()
        lib.call("rtfs_resid_proj_fwd", *args)
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    byt = 4.0 * B * (TF * (64 * 2 + 256 * 3 + 64) + 2 * T2 * F2 * 64)
    print(f"{lib.library_path()}: variant {variant} B {B}: median {1e3 * t[10]:.1f} us  min {1e3 * t[0]:.1f} us  {byt / (t[10] * 1e-3) / 1e12:.2f} TB/s algorithmic"
          f"   checksum {float(out.double().sum()):.10e} {float(py.double().abs().sum()):.10e}")
    if "residtime" in lib.library_path():
        tiles = (TF + 63) // 64
        per = min(128, (tiles * B + 255) // 256)
        raw = out.view(torch.int64).view(B, TF, 128)
        rows = []
        for gx in range((tiles + per - 1) // per):
            rows.append(raw[:, gx * per * 64: gx * per * 64 + 8, :7].cpu())  # [B][8 waves][7]
        v = torch.stack(rows, 1).double()  # [B][wg][wave][7]
        H = v[..., 6]
        for role, sl_, names in (("M", slice(0, 4), ("conv", "proj", "-", "-", "barrier", "head")), ("X", slice(4, 8), ("epilogue", "load_sv", "xform_e", "load_e", "barrier", "head"))):
            per_phase = (v[:, :, sl_, :6] / (H[:, :, sl_, None] + 2)).reshape(-1, 6)
            print(f"  {role} waves, cycles per phase (mean over waves): " + ", ".join(f"{n} {per_phase[:, j].mean():.0f}" for j, n in enumerate(names)) + f"   total {per_phase.sum(1).mean():.0f}")


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
