"""Debug aid: phase ablation of the LDS-DMA conv kernel (env CONV_DBG bitmask: 1 = no DMA, 2 = no MFMA, 4 = no barrier)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from vq_seg_amd import nnf

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = torch.device("cuda:0"); B = 32
conv = nn.Conv2d(1024, 256, 3, 1, 1, bias=False).to(dev); bn = nn.BatchNorm2d(256).to(dev).eval()
x = torch.randn(B, 512, 64, 64, device=dev).contiguous(memory_format=torch.channels_last).bfloat16()
x2 = torch.randn(B, 512, 64, 64, device=dev).contiguous(memory_format=torch.channels_last).bfloat16()
with torch.no_grad():
    t = timeit(lambda: nnf.conv_bn_act(x, conv, bn, x2=x2))
fl = 2.0 * B * 64 * 64 * 256 * 1024 * 9
print("CONV_DBG", os.environ.get("CONV_DBG"), f"{t*1e3:.1f} us {fl/t/1e9:.1f} TF/s")
