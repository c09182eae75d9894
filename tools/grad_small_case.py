import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
torch.set_num_threads(16)
from oracle.avnet_ref import avnet_forward
from oracle.regimes import stable_emb
from util import make_model, synth
NOGRAD = ("running_mean", "running_var", "scale_x", ".pe")
R,B,L,Tv,training = 1,3,2212,17,False
if len(sys.argv) > 1: L = int(sys.argv[1])
model, sd, cfg = make_model(R, "cuda")
mix, _, emb = synth.synth_inputs(B, L, Tv)
emb = stable_emb(sd, cfg, emb, training)
model.train(training)
if len(sys.argv) > 2: model.set_compute_dtype(sys.argv[2])
wgt = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(7))
import warnings; warnings.simplefilter("ignore")
out = model(mix.cuda(), emb.cuda()); (out * wgt.cuda()).sum().backward()
s = {k: (v.double().clone().requires_grad_(not k.endswith(NOGRAD)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
o64 = avnet_forward(s, cfg, mix.double(), emb.double(), training=training)
(o64 * wgt.double()).sum().backward()
print("forward rel", float((out.detach().cpu().double() - o64.detach()).norm() / o64.detach().norm()))
r64 = {k: v.grad for k, v in s.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
scale = max(float(g.norm()) for g in r64.values())
errs = []
for n, p in model.named_parameters():
    if float(r64[n].norm()) < 1e-6 * scale: continue
    errs.append((float((p.grad.double().cpu() - r64[n]).norm()) / (float(r64[n].norm()) + 1e-4 * scale), n, float(r64[n].norm()) / scale))
errs.sort(reverse=True)
for e in errs[:14]: print(f"{e[0]:.2e}  relnorm {e[2]:.1e}  {e[1]}")
for key in ("decoder.decoder.weight", "mask_generator.mask_generator.1.full_layer.2.weight", "mask_generator.mask_generator.0.weight",
            "refinement_module.audio_net.blocks.residual_conv.full_layer.2.weight", "refinement_module.audio_net.blocks.concat_layers.0.local_embedding.full_layer.2.weight",
            "refinement_module.audio_net.blocks.globalatt.2.attn_concat_proj.conv.weight", "refinement_module.audio_net.blocks.globalatt.1.linear.weight",
            "refinement_module.audio_net.blocks.globalatt.0.rnn.rnn_lst.0.weight", "refinement_module.audio_net.blocks.downsample_layers.0.full_layer.2.weight",
            "refinement_module.audio_net.blocks.projection.full_layer.2.weight", "refinement_module.crossmodal_fusion.fusion_module.audio_lstm.key_embed.full_layer.2.weight",
            "audio_bottleneck.full_layer.2.weight", "encoder.conv.full_layer.2.weight"):
    for e in errs:
        if e[1] == key: print(f"   {e[0]:.2e} {key}")
print("median", errs[len(errs)//2][0], "L", L)
