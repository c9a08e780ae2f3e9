"""Conditioning of the whole-model fixture: how far a tiny input perturbation moves logits and gradients on the CPU oracle.\n   usage: python tests/diagnostics/conditioning.py <size> <eps>"""
import sys; sys.path.insert(0,'/root/repo')
import torch
from oracle import torch_ref as R
from tests import cases, golden_io, synth
def rel(a,b): return ((a.double()-b.double()).abs().max()/(b.double().abs().max()+1e-12)).item()
fx = golden_io.load("model_v2")
base = synth.synth_state_dict(golden_io.layout("vqreptunet1x1"), fx.meta["model_seed"])
x, gt, scores = cases.model_inputs(b=2, s=int(sys.argv[1]))
ks=(0,0,512,512,512)
def run(xin):
    sd = {k:v.clone() for k,v in base.items()}
    with torch.no_grad():
        R.resnet_encoder(sd, xin, True, momentum=1.0)
        feats = R.resnet_encoder(sd, xin, False)[1:]
        for i in (2,3,4): sd[f"codebook.{i}.codebook.embedding.weight"] = cases.codebook_from_rows(cases.rows_of(feats[i]),512,900+i)
        R.vq_unet_forward(sd, xin, True, ks, gt=gt, version=2, margin=0.5, scale=30.0, momentum=1.0, th=0.7)
    p = {k:v.clone() for k,v in sd.items()}
    for k,v in p.items():
        if v.is_floating_point() and "running" not in k and "codebook" not in k: v.requires_grad_(True)
    logits, closs, usage, proto, aux = R.vq_unet_forward(p, xin, True, ks, gt=gt, version=2, margin=0.5, scale=30.0, th=0.7)
    total = torch.nn.functional.cross_entropy(logits, gt) + R.dice_loss(logits, gt) + 2.0*closs.sum()
    total.backward()
    return logits.detach(), {k:p[k].grad for k in ["segmentation_head.weight","decoder.blocks.4.1.0.weight","encoder.conv1.weight","decoder.blocks.0.0.0.weight"]}, aux
l0,g0,a0 = run(x)
noise = synth.uniform(4242, tuple(x.shape), -1.0, 1.0)*float(sys.argv[2])
l1,g1,a1 = run(x+noise)
print("logits", rel(l1,l0))
for k in g0: print(k, rel(g1[k], g0[k]))
print([ (a0['indices'][j]!=a1['indices'][j]).sum().item() for j in range(3)])
