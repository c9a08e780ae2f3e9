"""Where does the fp32 ("precise") weight-gradient error at bench scale come from?  One Conv3x3-BN-ReLU layer, 2048 -> 1024 channels on
32 x 16 x 16 pixels, GPU precise mode vs CPU float64, with train-mode and eval-mode BatchNorm; plus the raw weight-gradient entry point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from torch import nn
from tests import synth
from vq_seg_amd import nnf

torch.set_num_threads(16)
dev = torch.device("cuda:0")
cin, cout, b, s = 2048, 1024, 32, 16
x = synth.relu_features(3500, (b, cin, s, s))
w = synth.uniform(1, (cout, cin, 3, 3), -0.018, 0.018)
g = synth.uniform(3502, (b, cout, s, s), -1.0, 1.0)
gam, bet = synth.uniform(2, (cout,), 0.5, 1.5), synth.uniform(3, (cout,), -0.1, 0.1)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return f"max {((a - b).abs().max() / b.abs().max()).item():.2e} of scale, rel L2 {((a - b).norm() / b.norm()).item():.2e}"


for train in (True, False):
    conv, bn = nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(w), bn.weight.copy_(gam), bn.bias.copy_(bet)
        bn.running_mean.copy_(synth.uniform(4, (cout,), -0.1, 0.1)), bn.running_var.copy_(synth.uniform(5, (cout,), 0.5, 1.5))
    c64, b64 = nn.Conv2d(cin, cout, 3, padding=1, bias=False).double(), nn.BatchNorm2d(cout).double()
    c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    b64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    b64.train(train)
    xr = x.double().requires_grad_(True)
    y64 = F.relu(b64(c64(xr)))
    (y64 * g.double()).sum().backward()
    conv, bn = conv.to(dev), bn.to(dev)
    bn.train(train)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = nnf.conv_bn_act(xg, conv, bn)
    (y * g.to(dev)).sum().backward()
    print(f"BN train={train}: y {rel(y, y64)}; grad_x {rel(xg.grad, xr.grad)}; grad_w {rel(conv.weight.grad, c64.weight.grad)}; "
          f"grad_gamma {rel(bn.weight.grad, b64.weight.grad)}; grad_beta {rel(bn.bias.grad, b64.bias.grad)}")
    if train:
        # g_y the GPU's BN backward produced is not visible; rebuild the fp64 one and feed BOTH weight-gradient implementations with it
        yc = c64(x.double())
        gy64 = torch.autograd.grad(F.relu(b64(yc.requires_grad_(True))), yc, g.double())[0] if False else None
