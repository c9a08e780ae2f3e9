"""Times the kernels of one VQ forward (assign / gather) per shape with HIP events around repeated calls:
   python tools/bench_gather.py      (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
from tests import synth
dev = torch.device("cuda:0")
for n, c, k in [(131072, 512, 512), (32768, 1024, 512), (8192, 2048, 512)]:
    rows = synth.relu_features(1, (n, c)).to(dev).bfloat16()
    cb = synth.relu_features(2, (k, c)).to(dev)
    prep = _hip.vq_prepare(cb)
    idx = _hip.vq_assign(rows.float(), cb)
    for training in (True, False):
        for _ in range(3):
            _hip.vq_forward(rows, cb, training, 0.25, prepared=prep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _hip.vq_forward(rows, cb, training, 0.25, prepared=prep)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"N={n} C={c} K={k} training={training}: forward {dt * 1e6:8.1f} us  (rows {n * c * 2 / 1e6:.0f} MB bf16)", flush=True)
