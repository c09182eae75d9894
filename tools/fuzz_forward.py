"""Random-shape fuzz of the inference path against the oracle (24 shapes, fixed seed): python tools/fuzz_forward.py [f32|bf16x3|bf16]"""
import sys, random, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
torch.set_num_threads(16)
from util import make_model, rel, synth
from oracle.avnet_ref import avnet_forward
random.seed(1)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
tol = 2e-2 if dtype == "bf16" else 1e-3
worst = 0
for it in range(24):
    R = random.choice([1, 2, 3])
    B = random.choice([1, 1, 2, 3, 5, 7, 9])
    L = random.randint(1920, 26000)
    Tv = random.randint(1, max(2, L // 500))
    model, sd, cfg = make_model(R, "cuda")
    model.set_compute_dtype(dtype)
    mix, _, emb = synth.synth_inputs(B, L, Tv)
    with torch.no_grad():
        out = model(mix.cuda(), emb.cuda())
        ref = avnet_forward(sd, cfg, mix, emb)
    e = rel(out, ref)
    worst = max(worst, e)
    print(f"R {R} B {B} L {L} Tv {Tv}: {e:.2e}", flush=True)
    assert e < tol
print("worst", worst)
