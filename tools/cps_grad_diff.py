"""Per-parameter gradient of one CPS iteration: CPSTrainer (gradient sinks into flat buckets) vs plain autograd on the same models."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import cps_loop, cases
from tests.test_cps_parity_gpu import _trainer
import vq_seg_amd.models as models
from vq_seg_amd.loss import make_loss
from vq_seg_amd.measurement import Measurement
from vq_seg_amd.utils.lr_schedulers import CosineAnnealingLR

version = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tr, dev = _trainer(version, False)
l_in, l_tg, ul_in = [t.to(dev) for t in cps_loop.batches(1)[0]]
for o in tr.opts:
    for g in o.param_groups:
        g["lr"] = 0.0
tr.sched.start_lr = tr.sched.min_lr = 0.0
before = [p.detach().clone() for p in tr.models[0].parameters()]
tr.step(l_in, l_tg, ul_in)
moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, tr.models[0].parameters()))
print("params moved with lr=0:", moved)
g_sink = [{k: p.grad.detach().clone() for k, p in m.named_parameters()} for m in tr.models]

ns = types.SimpleNamespace(models=models, make_loss=make_loss, Measurement=Measurement, CosineAnnealingLR=CosineAnnealingLR)
pair = cps_loop.build_pair(ns, version, dev, prepare=lambda m, x, gt, v: cases.prepare_module_model(
    m, x, gt, v, to_input=lambda t: t.contiguous(memory_format=torch.channels_last)))
loop = cps_loop.Loop(ns, version, pair[0], pair[1], total_iters=1000, train=dict(learning_rate=0.0, min_lr=0.0))
loop.iteration(l_in, l_tg, ul_in)
worst = []
for mi, m in enumerate(pair):
    for k, p in m.named_parameters():
        a = g_sink[mi][k].double()
        if p.grad is None:
            if a.abs().max() > 0:
                worst.append((float("inf"), mi, k, "autograd None, sink nonzero"))
            continue
        b = p.grad.double()
        l2 = ((a - b).norm() / (b.norm() + 1e-30)).item()
        worst.append((l2, mi, k, f"|b|={b.norm().item():.3e}"))
worst.sort(reverse=True)
for w in worst[:25]:
    print(w)
print("median", worst[len(worst) // 2])
