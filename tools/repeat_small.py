import copy, sys, torch
sys.path.insert(0, "/root/repo")
from rtfs_net_amd import AVNet, synthetic
for (R,B,L,Tv) in ((3,10,18048,25),(6,32,32000,50)):
    cfg = synthetic.rtfs_audionet(R)
    model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict()))
    model = model.cuda()
    mix, _, emb = synthetic.synth_inputs(B, L, Tv)
    mix, emb = mix.cuda(), emb.cuda()
    for dtype in ("f32", "bf16x3", "bf16"):
        model.set_compute_dtype(dtype)
        with torch.no_grad():
            ref = model(mix, emb).double()
            diffs = []
            for it in range(150):
                if it % 10 == 0:  # disturb the allocator / caches like other tests do
                    junk = [torch.randn(1 << 22, device="cuda") for _ in range(8)]; del junk
                    if R == 3:
                        model._hip.variants["resid"] = (it // 10) % 3 + 1
                        model(mix, emb)
                        model._hip.variants["resid"] = 0
                out = model(mix, emb).double()
                d = float((out - ref).norm() / ref.norm())
                if d > 0: diffs.append(d)
        print(f"R={R} B={B} {dtype}: {len(diffs)} of 150 runs differ from the first; max rel {max(diffs) if diffs else 0:.2e}", flush=True)
