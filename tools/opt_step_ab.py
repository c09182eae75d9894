import sys, copy, torch, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from util import make_model
model, _, _ = make_model(6, "cuda")
for p in model.parameters(): p.grad = torch.randn_like(p)
for kw in ({}, {"fused": True}):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.1, **kw)
    for _ in range(5):
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); opt.step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); opt.step()
    b.record(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); opt.step()
    torch.cuda.synchronize()
    print(kw, "gpu ms per (clip+step):", a.elapsed_time(b) / 20, " wall:", (time.time() - t0) / 20 * 1e3)
from rtfs_net_amd.optim import FusedAdamW
opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
for _ in range(5):
    opt.step(max_norm=5.0)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    opt.step(max_norm=5.0)
b.record(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    opt.step(max_norm=5.0)
torch.cuda.synchronize()
print("FusedAdamW.step(max_norm=5.0) gpu ms:", a.elapsed_time(b) / 20, " wall:", (time.time() - t0) / 20 * 1e3)
