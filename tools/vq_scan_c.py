"""vq_assign time against the channel count at fixed N, K: slope = main-loop rate, intercept = per-launch fixed cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
dev = torch.device("cuda:0")
n, k = 131072, 512
L = _hip.lib()
for c in (16, 32, 64, 128, 256, 512, 1024, 2048):
    x = torch.relu(torch.randn(n, c, device=dev)); W = torch.relu(torch.randn(k, c, device=dev))
    prep = _hip.vq_prepare(W)
    for _ in range(3): _hip.vq_assign(x, W, prepared=prep)
    L.vqseg_profile_begin(64)
    for _ in range(10): _hip.vq_assign(x, W, prepared=prep)
    recs = _hip.profile_collect(64)
    ms = sum(r[3] for r in recs) / len(recs)
    print(f"C={c:5d}  kernel {ms * 1e3:8.1f} us   {2.0 * n * k * c / ms / 1e9:7.1f} TF/s", flush=True)
