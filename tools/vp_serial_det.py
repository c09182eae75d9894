import copy, sys, torch
sys.path.insert(0, "/root/repo")
from rtfs_net_amd import AVNet, synthetic
cfg = synthetic.rtfs_audionet(6)
model = AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
model.load_state_dict(synthetic.synth_state_dict(model.state_dict()))
model = model.cuda()
mix, _, emb = synthetic.synth_inputs(32, 32000, 50)
mix, emb = mix.cuda(), emb.cuda()
model.set_compute_dtype("bf16")
def count(n=30):
    with torch.no_grad():
        ref = model(mix, emb); bad = 0
        for _ in range(n): bad += int(not torch.equal(model(mix, emb), ref))
    return bad
print("side stream:", count(), flush=True)
model._hip._vp_stream = torch.cuda.current_stream()
print("video branch on the main stream:", count(), flush=True)
