"""Experiment: one forward of batch 32 against two concurrent forwards of batch 16 on two streams (two model instances, same weights): do MFMA-bound and
HBM-bound kernels of the two halves overlap enough to pay for the smaller launches?"""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from util import make_model, synth  # noqa: E402

B = 32
m1, _, _ = make_model(6, "cuda")
m2, _, _ = make_model(6, "cuda")
mix, _, emb = synth.synth_inputs(B, 32000, 50)
mix, emb = mix.cuda(), emb.cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def whole():
    return m1(mix, emb)


def halves():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur), s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = m1(mix[:B // 2], emb[:B // 2])
    with torch.cuda.stream(s2):
        b = m2(mix[B // 2:], emb[B // 2:])
    cur.wait_stream(s1), cur.wait_stream(s2)
    return a, b


with torch.no_grad():
    for name, fn in (("batch 32", whole), ("2 x 16 on two streams", halves), ("batch 32", whole), ("2 x 16 on two streams", halves)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms")
    a, b = halves()
    w = whole()
    torch.cuda.synchronize()
    print("max diff", float((torch.cat([a, b]) - w).abs().max()))
