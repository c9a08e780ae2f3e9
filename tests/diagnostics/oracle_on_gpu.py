"""Debug: run the oracle's functional model with torch ops ON THE GPU and compare grads with golden."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import torch_ref as R
from tests import cases, golden_io, synth
dev = torch.device(sys.argv[1] if len(sys.argv) > 1 else "cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
version = 2
fx = golden_io.load(f"model_v{version}")
sd = {k: v.to(dev) for k, v in synth.synth_state_dict(golden_io.layout("vqreptunet1x1"), fx.meta["model_seed"]).items()}
x, gt, scores = [t.to(dev) for t in cases.model_inputs()]
ks = (0, 0, 512, 512, 512)
with torch.no_grad():
    R.resnet_encoder(sd, x, True, momentum=1.0)
    feats = R.resnet_encoder(sd, x, False)[1:]
    for i in (2, 3, 4):
        sd[f"codebook.{i}.codebook.embedding.weight"] = cases.codebook_from_rows(cases.rows_of(feats[i]), 512, 900 + i)
    R.vq_unet_forward(sd, x, True, ks, gt=gt, version=version, margin=0.5, scale=30.0, momentum=1.0, th=0.7)
p = {k: v.clone() for k, v in sd.items()}
for k, v in p.items():
    if v.is_floating_point() and "running" not in k and "codebook" not in k:
        v.requires_grad_(True)
logits, closs, usage, proto, aux = R.vq_unet_forward(p, x, True, ks, gt=gt, version=version, margin=0.5, scale=30.0, th=0.7)
print("train logits", rel(logits, fx["train_logits"]))
total = (logits * cases.logits_cotangent(logits.shape).to(dev)).sum() + fx.meta["loss_scale"] * closs.sum()
total.backward()
for key in [k[5:] for k in fx if k.startswith("grad/")]:
    print("   grad", key, rel(golden_io.probe(p[key].grad), fx["grad/" + key]))
