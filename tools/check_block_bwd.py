"""GPU diagnostic: the dual-path and attention adjoints IN ISOLATION against float64 autograd of the oracle's stage functions,
at full-size compressed shapes (T2 = 125: time sequences of 118 windows = two 64-row tiles per sequence).

    python tools/check_block_bwd.py [T2] [B]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import make_model, rel  # noqa: E402

from oracle.avnet_ref import P, dual_path_rnn, mhsa2d  # noqa: E402
from rtfs_net_amd.models.hip_train import Ctx, HipTrainer, grads_to_reference  # noqa: E402


def main(T2=125, B=1):
    model, sd, cfg = make_model(1, "cuda")
    tr = HipTrainer(model)
    pw = tr.weights()
    bw = pw.blocks[0]
    g = torch.Generator().manual_seed(5)
    G0 = torch.randn(B, T2, 64, 64, generator=g)       # channels-last [B][T2][F2][64]
    dOut = torch.randn(B, T2, 64, 64, generator=g)
    pre = "refinement_module.audio_net.blocks.globalatt."
    nograd = ("scale_x",)
    for name, dim, key in (("freq", 4, "dp0"), ("time", 3, "dp1")):
        idx = 0 if dim == 4 else 1
        Gd = G0.cuda().reshape(-1).clone()
        sv = Ctx()
        tr._dual_path_fwd(Gd, bw[key], B, T2, dim, sv)
        out_hip = Gd.view(B, T2, 64, 64).cpu()
        dG = dOut.cuda().reshape(-1).clone()
        gr = {}
        tr._dual_path_bwd(dG, bw[key], sv, B, T2, dim, gr, f"blk.{key}")
        torch.cuda.synchronize()
        # oracle in float64 (NCHW [B, 64, T2, F2])
        sd64 = {k: (v.double().clone().requires_grad_(not k.endswith(nograd))) for k, v in sd.items() if k.startswith(f"{pre}{idx}.")}
        x = G0.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        y = dual_path_rnn(x, P(sd64, f"{pre}{idx}."), dim=dim, hid=32)
        (y * dOut.double().permute(0, 3, 1, 2)).sum().backward()
        print(f"[{name}] forward rel {rel(out_hip.permute(0, 3, 1, 2), y.detach()):.3e}   d(input) rel {rel(dG.view(B, T2, 64, 64).cpu().permute(0, 3, 1, 2), x.grad):.3e}")
        k = f"blk.{key}."
        q = f"{pre}{idx}."
        got = {q + "norm.gamma": gr[k + "g"].view(1, 64, 1, 1), q + "norm.beta": gr[k + "b"].view(1, 64, 1, 1),
               q + "rnn.rnn_lst.0.weight": gr[k + "w0"].view(256, 8, 64).permute(2, 1, 0).reshape(512, 256),
               q + "linear.weight": gr[k + "ct_w"].view(64, 8, 64).permute(2, 0, 1).flip(2), q + "linear.bias": gr[k + "ct_b"]}
        for l in range(4):
            got[q + f"rnn.rnn_lst.{l}.weight_c"] = gr[k + f"l{l}.wc"]
            got[q + f"rnn.rnn_lst.{l}.bias"] = gr[k + f"l{l}.bias"]
            if l > 0:
                got[q + f"rnn.rnn_lst.{l}.weight"] = gr[k + f"l{l}.w"].view(3, 64, 64).permute(2, 1, 0).reshape(64, 192)
        for n, v in got.items():
            print(f"    {n[len(pre):]:40s} rel {rel(v.cpu().reshape(sd64[n].shape), sd64[n].grad):.3e}")
    # attention
    a = bw["attn"]
    G = G0.cuda().reshape(-1).clone()
    k = Ctx()
    dev = G.device
    from rtfs_net_amd import lib

    k.G2 = G.clone()
    k.Q = torch.empty(B * 4 * T2 * 256, device=dev)
    k.K = torch.empty_like(k.Q)
    k.V = torch.empty(B * 4 * T2 * 1024, device=dev)
    k.Ypre96 = torch.empty(B * T2 * 64 * 96, device=dev)
    lib.call("rtfs_attn_qkv_fwd", G, a["w"], a["bias"], a["slope"], a["gq"], a["bq"], a["gk"], a["bk"], a["gv"], a["bv"], k.Q, k.K, k.V, k.Ypre96, B, T2)
    k.O = torch.empty(B * T2 * 4096, device=dev)
    k.LSE = torch.empty(B * 4 * T2, device=dev)
    lib.call("rtfs_attn_core_fwd", k.Q, k.K, k.V, k.O, k.LSE, B, T2)
    k.Ypre_o = torch.empty(B * T2 * 4096, device=dev)
    lib.call("rtfs_attn_out_fwd", k.O, a["ow"], a["ob"], a["oslope"], a["og"], a["obe"], G, k.Ypre_o, B, T2)
    dG = dOut.cuda().reshape(-1).clone()
    gr = {}
    tr._attn_bwd(dG, a, k, B, T2, gr, "blk.attn")
    torch.cuda.synchronize()
    sd64 = {kk: v.double().clone().requires_grad_(True) for kk, v in sd.items() if kk.startswith(f"{pre}2.")}
    x = G0.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = mhsa2d(x, P(sd64, f"{pre}2."), 4)
    (y * dOut.double().permute(0, 3, 1, 2)).sum().backward()
    print(f"[attn] forward rel {rel(G.view(B, T2, 64, 64).cpu().permute(0, 3, 1, 2), y.detach()):.3e}   d(input) rel {rel(dG.view(B, T2, 64, 64).cpu().permute(0, 3, 1, 2), x.grad):.3e}")
    print("    dW(qkv) norm", float(gr["blk.attn.w"].norm()), " ref Values.0.conv.weight grad norm", float(sd64[f"{pre}2.Values.0.conv.weight"].grad.norm()))
    wq = gr["blk.attn.w"].view(96, 64).cpu()
    off = 0
    for name, nch in (("Queries", 4), ("Keys", 4), ("Values", 16)):
        for h in range(4):
            r = sd64[f"{pre}2.{name}.{h}.conv.weight"].grad.reshape(nch, 64)
            print(f"    {name}.{h}.conv.weight rel {rel(wq[off:off + nch], r):.3e}")
            off += nch
    print(f"    attn_concat_proj.conv.weight rel {rel(gr['blk.attn.ow'].view(64, 64).cpu(), sd64[f'{pre}2.attn_concat_proj.conv.weight'].grad.reshape(64, 64)):.3e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 125, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
