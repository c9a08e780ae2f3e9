"""Times the stem's patch-matrix kernel at the bench's shape (32 x 512 x 512 x 3 fp32 -> [2.1 M rows][160] bf16 / split-3): HIP events, 20 launches.
    python tools/bench_im2col.py            (VQSEG_LIB / VQSEG_OPTS select the build / the gather kernel: im2col_strip=0)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
dev = torch.device("cuda:0")
L = _hip.lib()
n, h, w, kp = 32, 512, 512, 160
ho = wo = 256
x = torch.rand(n, h, w, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
for kind, name, width, dt in ((1, "bf16", kp, torch.bfloat16), (2, "split-3", 2 * kp, torch.bfloat16), (0, "fp32", kp, torch.float32)):
    out = torch.empty(n * ho * wo, width, dtype=dt, device=dev)
    for reflect in (1, 0):
        for _ in range(3):
            assert L.vqseg_im2col_f(kind, x.data_ptr(), n, h, w, 3, 7, 7, 2, 3, reflect, ho, wo, kp, out.data_ptr(), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.vqseg_im2col_f(kind, x.data_ptr(), n, h, w, 3, 7, 7, 2, 3, reflect, ho, wo, kp, out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        mb = (out.numel() * out.element_size() + x.numel() * 4) / 1e6
        print(f"{name:8s} reflect={reflect}: {us:7.1f} us  {mb / us:5.2f} TB/s ({mb:.0f} MB)", flush=True)
