import copy, os, sys, torch
sys.path.insert(0, "/root/repo")
from oracle import synth
from rtfs_net_amd import AVNet
B, R, L, Tv = 32, 6, 32000, 50
dev = torch.device("cuda:0")
model = AVNet(print_macs=False, **copy.deepcopy(synth.rtfs_audionet(R)))
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.to(dev).train(True)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
mix, _, emb = synth.synth_inputs(B, L, Tv)
mix, emb = mix.to(dev), emb.to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    model(mix, emb).square().mean().backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=14)
LEAF = ("aten::copy_", "aten::mul", "aten::fill_", "aten::div", "aten::sub", "aten::cat", "aten::add", "aten::add_", "aten::neg", "aten::flip", "aten::clamp_min", "aten::mul_", "aten::stack")
rows = [(e.count, e.device_time_total, e.key, e.stack) for e in ka if e.key in LEAF and e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
tot = 0
for c, t, k, st in rows[:60]:
    tot += t
    s = [x for x in st if "rtfs_net_amd" in x or "train_glue" in x or "optim" in x or "clip_grad" in x]
    print(f"{c:5d} {t:9.1f} us  {k:16s} {' <- '.join(x.split('/')[-1][:70] for x in s[:2])}")
print("sum", sum(r[1] for r in rows))
