"""Debug aid: per-quantity errors of conv_bn_act (HIP) vs the CPU fp32 reference for the test cases."""
import copy, os, sys
import torch, torch.nn.functional as F
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth
from tests.test_nn_gpu import CONV_CASES, cl, dev, rel
from vq_seg_amd import nnf

def run(case, training, mode):
    n, cin, cout, h, w, k, s, p, reflect, c2, use_res = case
    seed = sum(case[:8]) + 7 * int(training)
    conv = nn.Conv2d(cin + c2, cout, k, s, p, bias=False, padding_mode="reflect" if reflect else "zeros")
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(synth.uniform(seed, tuple(conv.weight.shape), -1, 1) * (2.0 / ((cin + c2) * k * k)) ** 0.5)
        bn.weight.copy_(synth.uniform(seed + 1, (cout,), 0.5, 1.5)); bn.bias.copy_(synth.uniform(seed + 2, (cout,), -0.3, 0.3))
        bn.running_mean.copy_(synth.uniform(seed + 3, (cout,), -0.2, 0.2)); bn.running_var.copy_(synth.uniform(seed + 4, (cout,), 0.5, 1.5))
    conv.train(training), bn.train(training)
    dt = torch.float32 if mode == "precise" else torch.bfloat16
    q = lambda t: t.to(dt).float()
    x = q(synth.uniform(seed + 5, (n, cin, h, w), -1, 1)); x2 = q(synth.uniform(seed + 6, (n, c2, h, w), -1, 1)) if c2 else None
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    res = q(synth.uniform(seed + 7, (n, cout, ho, wo), -1, 1)) if use_res else None
    g = q(synth.uniform(seed + 8, (n, cout, ho, wo), -1, 1))
    conv_g, bn_g = copy.deepcopy(conv).to(dev()), copy.deepcopy(bn).to(dev())
    xr = x.clone().requires_grad_(True); x2r = x2.clone().requires_grad_(True) if c2 else None; rr = res.clone().requires_grad_(True) if use_res else None
    y = bn(conv(torch.cat((xr, x2r), 1) if c2 else xr))
    if use_res: y = y + rr
    y = F.relu(y); y.backward(g)
    xg = cl(x).to(dt).requires_grad_(True); x2g = cl(x2).to(dt).requires_grad_(True) if c2 else None; rg = cl(res).to(dt).requires_grad_(True) if use_res else None
    out = nnf.conv_bn_act(xg, conv_g, bn_g, relu=True, residual=rg, x2=x2g)
    out.backward(cl(g).to(dt))
    e = dict(y=rel(out.float(), y), gx=rel(xg.grad.float(), xr.grad), gw=rel(conv_g.weight.grad, conv.weight.grad), gg=rel(bn_g.weight.grad, bn.weight.grad), gb=rel(bn_g.bias.grad, bn.bias.grad))
    if c2: e["gx2"] = rel(x2g.grad.float(), x2r.grad)
    if use_res: e["gr"] = rel(rg.grad.float(), rr.grad)
    print(mode, "train" if training else "eval ", case, {k: f"{v:.1e}" for k, v in e.items()})

for mode in ("precise", "fast"):
    for case in CONV_CASES:
        if mode == "fast" and (case[1] % 8 or case[9] % 8 or case[2] % 8): continue
        for tr in (True, False):
            run(case, tr, mode)
