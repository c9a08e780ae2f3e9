import os, sys, torch
sys.path.insert(0, "/root/repo")
from torch import nn
from vq_seg_amd import nnf
dev = torch.device("cuda:0")
conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False, padding_mode="reflect").to(dev); bn = nn.BatchNorm2d(64).to(dev)
x = torch.rand(32, 3, 512, 512, device=dev).contiguous(memory_format=torch.channels_last)
def run():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return nnf.stem_conv_bn_act(x, conv, bn)
for _ in range(3): run()
import vq_seg_amd._hip as _hip
L = _hip.lib()
_hip.conv_profile_begin(64)
for _ in range(10): run()
torch.cuda.synchronize()
recs = _hip.conv_profile_collect(64)
ms = sorted(r[2] for r in recs)
print(os.environ.get("VQSEG_LIB", "in-tree")[-20:], f"stem conv launches {len(recs)}: median {1e3 * ms[len(ms) // 2]:.1f} us")
