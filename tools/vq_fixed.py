"""Fixed cost of the distance+argmin kernel: time against N at tiny C (main loop ~ nothing) and against C at fixed N."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
dev = torch.device("cuda:0")
L = _hip.lib()
def t(n, c, k, dt=torch.bfloat16):
    x = torch.relu(torch.randn(n, c, device=dev)).to(dt); W = torch.relu(torch.randn(k, c, device=dev))
    prep = _hip.vq_prepare(W)
    for _ in range(5): _hip.vq_assign(x, W, prepared=prep)
    L.vqseg_profile_begin(64)
    for _ in range(20): _hip.vq_assign(x, W, prepared=prep)
    recs = _hip.profile_collect(64)
    return sorted(r[3] for r in recs)[len(recs) // 2] * 1e3
for n in (128, 16384, 32768, 65536, 131072, 262144):
    print("N", n, " ".join(f"C{c}:{t(n, c, 512):7.1f}" for c in (16, 64, 256, 512)), flush=True)
