"""Micro-benchmark of the conv kernels (forward / dgrad / wgrad) on representative layers of the model at batch B."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from vq_seg_amd import nnf, _hip

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = torch.bfloat16 if (len(sys.argv) <= 2 or sys.argv[2] == "bf16") else torch.float32
dev = torch.device("cuda:0")
LAYERS = [  # name, cin, cout, hw, k, stride, reflect, c2
    ("l1.conv1 1x1 256->64 @128", 256, 64, 128, 1, 1, False, 0),
    ("l1.conv2 3x3 64->64 @128 r", 64, 64, 128, 3, 1, True, 0),
    ("l1.conv3 1x1 64->256 @128", 64, 256, 128, 1, 1, False, 0),
    ("l2.conv2 3x3 128->128 @64 r", 128, 128, 64, 3, 1, True, 0),
    ("l3.conv2 3x3 256->256 @32 r", 256, 256, 32, 3, 1, True, 0),
    ("l3.conv3 1x1 256->1024 @32", 256, 1024, 32, 1, 1, False, 0),
    ("l4.conv2 3x3 512->512 @16 r", 512, 512, 16, 3, 1, True, 0),
    ("dec0.0 3x3 2048->1024 @16", 2048, 1024, 16, 3, 1, False, 0),
    ("dec1.0 3x3 1024+1024->512 @32", 1024, 512, 32, 3, 1, False, 1024),
    ("dec2.0 3x3 512+512->256 @64", 512, 256, 64, 3, 1, False, 512),
    ("dec3.0 3x3 256+256->128 @128", 256, 128, 128, 3, 1, False, 256),
    ("dec3.1 3x3 128->128 @128", 128, 128, 128, 3, 1, False, 0),
    ("dec4.0 3x3 128+64->32 @256", 128, 32, 256, 3, 1, False, 64),
    ("dec4.1 3x3 32->32 @256", 32, 32, 256, 3, 1, False, 0),
]
print(f"B={B} dtype={dt}")
for name, cin, cout, hw, k, s, refl, c2 in LAYERS:
    conv = nn.Conv2d(cin + c2, cout, k, s, k // 2, bias=False, padding_mode="reflect" if refl else "zeros").to(dev)
    bn = nn.BatchNorm2d(cout).to(dev)
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).to(dt).requires_grad_(True)
    x2 = torch.randn(B, c2, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).to(dt).requires_grad_(True) if c2 else None
    flops = 2.0 * B * hw * hw / (s * s) * cout * (cin + c2) * k * k
    def fwd():
        return nnf.conv_bn_act(x, conv, bn, x2=x2)
    t_f = timeit(fwd)
    out = fwd(); g = torch.randn_like(out)
    def fb():
        o = fwd(); o.backward(g)
    t_fb = timeit(fb)
    print(f"{name:34s} {flops/1e9:8.1f} GF  fwd(conv+bn) {t_f*1e3:8.1f} us {flops/t_f/1e9:7.1f} TF/s | fwd+bwd {t_fb*1e3:8.1f} us  {3*flops/t_fb/1e9:7.1f} TF/s")
