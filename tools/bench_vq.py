"""Micro-benchmark of the VQ kernels on one MI355X (not the headline bench; see bench.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    print(torch.cuda.get_device_name(0), "B =", B)
    for name, n, c, k in [("L2 512^2", B * 4096, 512, 512), ("L3 512^2", B * 1024, 1024, 512),
                          ("L4 512^2", B * 256, 2048, 512), ("L2 K=256", B * 4096, 512, 256),
                          ("L2 1024^2 K=1024", B * 16384 // 4, 512, 1024)]:
        x = torch.relu(torch.randn(n, c, device=dev))
        W = torch.relu(torch.randn(k, c, device=dev))
        prep = _hip.vq_prepare(W)
        t_as = timeit(lambda: _hip.vq_assign(x, W, prepared=prep))
        t_fw = timeit(lambda: _hip.vq_forward(x, W, True, 1.0, prepared=prep))
        flops = 2.0 * n * k * c
        byts = n * (c * 4 * 2 + 8)
        print(f"{name:18s} N={n:7d} C={c:4d} K={k:4d}  assign {t_as*1e3:8.1f} us  {flops/t_as/1e9:7.1f} TF/s "
              f"({flops/t_as/1e9/157.3*100:4.1f}% of 157.3)   forward {t_fw*1e3:8.1f} us  alg {byts/t_fw/1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
