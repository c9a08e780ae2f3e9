"""Kernel time (in-stream HIP events, vqseg_profile_*) of the distance+argmin kernel on the three bench shapes (B = 32 at 512^2,
bf16 rows as in the bench), for A/B runs of library builds: VQSEG_LIB=/path/other.so python tools/vq_shapes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
dev = torch.device("cuda:0")
L = _hip.lib()
tot_f = tot_t = 0.0
out = []
for n, c, k in ((131072, 512, 512), (32768, 1024, 512), (8192, 2048, 512)):
    x = torch.relu(torch.randn(n, c, device=dev)).to(torch.bfloat16); W = torch.relu(torch.randn(k, c, device=dev))
    prep = _hip.vq_prepare(W)
    for _ in range(5): _hip.vq_assign(x, W, prepared=prep)
    L.vqseg_profile_begin(64)
    for _ in range(20): _hip.vq_assign(x, W, prepared=prep)
    recs = _hip.profile_collect(64)
    ms = sorted(r[3] for r in recs)[len(recs) // 2]
    tot_f += 2.0 * n * k * c; tot_t += ms
    out.append(f"N{n}xC{c}: {ms * 1e3:7.1f} us {2.0 * n * k * c / ms / 1e9:6.1f} TF")
print(os.environ.get("VQSEG_LIB", "in-tree")[-30:], " | ".join(out), f"| all: {tot_t * 1e3:7.1f} us {tot_f / tot_t / 1e9:6.1f} TF = {tot_f / tot_t / 1e9 / 157.3:.3f}", flush=True)
