"""Experiment: capture the whole inference forward in one hipGraph (torch.cuda.CUDAGraph) and compare latency with eager launches."""
import copy
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from oracle import synth  # noqa: E402
from rtfs_net_amd import AVNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = 6
L, Tv = 32000, 50
dev = torch.device("cuda:0")
model = AVNet(print_macs=False, **copy.deepcopy(synth.rtfs_audionet(R))).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.to(dev)
mix, _, emb = synth.synth_inputs(B, L, Tv)
mix, emb = mix.to(dev), emb.to(dev)


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    ref = model(mix, emb).clone()
    eager = bench(lambda: model(mix, emb))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(mix, emb)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model(mix, emb)
    g.replay()
    torch.cuda.synchronize()
    print("graph vs eager max diff", float((out - ref).abs().max()))
    graph = bench(g.replay)
print(f"B={B}: eager {eager:.3f} ms, graph {graph:.3f} ms")
