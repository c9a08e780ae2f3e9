"""The convolution layers that drag `roofline_conv` down, one at a time at the bench's shapes (B = 32, bf16, training
mode: forward + backward through nnf.conv_bn_act with gradient sinks), for rocprofv3 counter passes (tools/pmc_weak.sh):
    python tools/weak_layers.py <layer> [reps]
layers: l1.conv2 (64 -> 64 3x3 reflect @128^2), l2.0.conv2 / l3.0.conv2 / l4.0.conv2 (stride-2 3x3 reflect), dec4.0 (128 + 64 -> 32 @256^2),
dec4.1 (32 -> 32 @256^2)."""
import os, sys
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import nnf
dev = torch.device("cuda:0")
LAYERS = {  # cin, c2 (concat), cout, k, stride, reflect, H
    "l1.conv2": (64, 0, 64, 3, 1, True, 128), "l2.0.conv2": (128, 0, 128, 3, 2, True, 128), "l3.0.conv2": (256, 0, 256, 3, 2, True, 64),
    "l4.0.conv2": (512, 0, 512, 3, 2, True, 32), "dec4.0": (128, 64, 32, 3, 1, False, 256), "dec4.1": (32, 0, 32, 3, 1, False, 256),
    "stem": (3, 0, 64, 7, 2, True, 512), "l1.0.conv1": (64, 0, 64, 1, 1, True, 128),
}
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cin, c2, cout, k, stride, reflect, H = LAYERS[name]
B = 32
torch.manual_seed(0)
conv = nn.Conv2d(cin + c2, cout, k, stride, k // 2, bias=False, padding_mode="reflect" if reflect else "zeros").to(dev)
bn = nn.BatchNorm2d(cout).to(dev)
for p in list(conv.parameters()) + list(bn.parameters()):
    p.grad = torch.zeros_like(p)
    p._vq_grad_sink = None
x = torch.relu(torch.randn(B, cin, H, H, device=dev)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
if name == "stem":
    x = torch.rand(B, 3, H, H, device=dev).contiguous(memory_format=torch.channels_last)
x2 = torch.relu(torch.randn(B, c2, H, H, device=dev)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) if c2 else None
for i in range(reps + 1):
    if name == "stem":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = nnf.stem_conv_bn_act(x, conv, bn)
    else:
        y = nnf.conv_bn_act(x, conv, bn, x2=x2)
    y.float().mean().backward()
    x.grad = None
    if x2 is not None:
        x2.grad = None
torch.cuda.synchronize()
flops = 2.0 * k * k * (cin + c2) * cout * B * (H // stride) ** 2
print(f"{name}: {cin}+{c2} -> {cout} k{k} s{stride} {'reflect' if reflect else 'zeros'} @{H}^2 B{B}: {flops / 1e9:.1f} GFLOP per pass; "
      f"algorithmic MB: x {B * H * H * (cin + c2) * 2 / 1e6:.1f}, y {B * (H // stride) ** 2 * cout * 2 / 1e6:.1f}, w {k * k * (cin + c2) * cout * 2 / 1e6:.2f}")
