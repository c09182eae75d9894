"""GPU: run the batch-32 forward repeatedly and report the first kernel launch whose tensor arguments differ bitwise from the first run
(every lib.call is followed by a checksum of all its tensor arguments; the first launch that differs is the non-deterministic one)."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtfs_net_amd import AVNet, lib, synthetic  # noqa: E402

R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = synthetic.rtfs_audionet(R)
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synthetic.synth_state_dict(model.state_dict()))
model = model.cuda()
mix, _, emb = synthetic.synth_inputs(32, 32000, 50)
mix, emb = mix.cuda(), emb.cuda()
orig_call = lib.call
log = []


def spy(name, *args):
    orig_call(name, *args)
    sums = []
    for a in args:
        ts = a if isinstance(a, (list, tuple)) else [a]
        for t in ts:
            if isinstance(t, torch.Tensor) and t.numel() > 0:
                v = t.detach().contiguous().view(-1)
                v = v.view(torch.int64) if (v.element_size() == 8) else (v[: v.numel() // 2 * 2].view(torch.int32) if v.element_size() == 4 else v.view(torch.int16))
                sums.append(v.to(torch.int64).sum())
    log.append((name, torch.stack(sums) if sums else None))


lib.call = spy
import rtfs_net_amd.models.hip_path as hp  # noqa: E402

hp.lib.call = spy
for dtype in (sys.argv[3:] or ["f32", "bf16x3"]):
    model.set_compute_dtype(dtype)
    ref, found = None, {}
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        log.clear()
        with torch.no_grad():
            model(mix, emb)
        torch.cuda.synchronize()
        cur = [(n, None if s is None else s.cpu()) for n, s in log]
        if ref is None:
            ref = cur
            continue
        for i, ((n, s), (n0, s0)) in enumerate(zip(cur, ref)):
            if s is not None and not torch.equal(s, s0):
                key = (i, n, tuple((s != s0).nonzero().flatten().tolist()))  # (launch index, entry point, indices of the differing tensor arguments)
                found[key] = found.get(key, 0) + 1
                break
        if os.environ.get("ND_CHAIN", "0") == "1":
            ref = cur  # compare consecutive runs instead of every run with the first
    print(dtype, "first differing launch per run (launch index, entry point): count ->", found or "none")
