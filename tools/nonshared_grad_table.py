import sys, copy, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from util import load_npz, synth, rel
from oracle.regimes import NONSHARED_CASES, GRAD_WEIGHT_SEED, case_name
from rtfs_net_amd import AVNet
training,B,L,R,Tv=NONSHARED_CASES[0]
cfg=synth.rtfs_audionet(R); cfg["audio_params"]["shared"]=False
model=AVNet(print_macs=False, **copy.deepcopy(cfg)).eval()
sd=synth.synth_state_dict(model.state_dict()); model.load_state_dict(sd); model=model.cuda()
z=load_npz(case_name("nonshared",training,B,L,R,Tv)+".npz")
mix,_,_=synth.synth_inputs(B,L,Tv); emb=torch.from_numpy(z["emb"])
wgt=torch.randn(B,1,L,generator=torch.Generator().manual_seed(GRAD_WEIGHT_SEED))
out=model(mix.cuda(),emb.cuda()); (out*wgt.cuda()).sum().backward()
ref={k[5:]:torch.from_numpy(z[k]).double() for k in z.files if k.startswith("grad.")}
scale=max(float(g.norm()) for g in ref.values())
rows=[]
for n,p in model.named_parameters():
    if float(ref[n].norm())<1e-6*scale: continue
    rows.append((float((p.grad.double().cpu()-ref[n]).norm())/(float(ref[n].norm())+1e-4*scale), n))
rows.sort(reverse=True)
for e,n in rows[:25]: print(f"{e:.2e} {n}")
print("median", rows[len(rows)//2][0])
