"""Print per-tensor errors of the GPU model vs the golden fixtures (debug aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases, golden_io, synth
from tests.test_model_gpu import build, dev

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()

for version in (1, 2):
    fx = golden_io.load(f"model_v{version}")
    model = build(fx.meta["name"], fx.meta["margin"], fx.meta["scale"], fx.meta["model_seed"])
    x, gt, scores = [t.to(dev()) for t in cases.model_inputs()]
    model.eval()
    with torch.no_grad():
        logits = model(x)[0]
        l2 = model(x)[0]
    print(version, "eval logits rel", rel(logits, fx["eval_logits"]), "rerun diff", rel(l2, logits))
    model.train()
    kw = dict(percent=fx.meta["percent"]) if version == 1 else dict(th=fx.meta["th"])
    logits, closs, usage, proto = model(x, gt, **kw)
    print("  train logits", rel(logits, fx["train_logits"]), "closs", rel(closs, fx["train_loss"]), "proto", rel(proto, fx["train_proto"]), usage.tolist(), fx["train_usage"].tolist())
    total = (logits * cases.logits_cotangent(logits.shape).to(dev())).sum() + fx.meta["loss_scale"] * closs.sum()
    if version == 1:
        total = total + fx.meta["proto_scale"] * proto
    total.backward()
    named = dict(model.named_parameters())
    for key in [k[5:] for k in fx if k.startswith("grad/")]:
        print("   grad", key, rel(golden_io.probe(named[key].grad), fx["grad/" + key]))
