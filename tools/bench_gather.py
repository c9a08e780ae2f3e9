"""vq_gather_kernel per shape: forward minus assign (in-stream events), bf16 and f32 rows, healthy and collapsed code usage."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for dt in (torch.bfloat16, torch.float32):
    for n, c, k in ((131072, 512, 512), (32768, 1024, 512), (8192, 2048, 512)):
        for usage in ("kmeans-like", "collapsed"):
            x = torch.relu(torch.randn(n, c, device=dev)).to(dt)
            W = x[torch.randperm(n, device=dev)[:k]].float().contiguous() if usage == "kmeans-like" else torch.randn(k, c, device=dev)
            prep = _hip.vq_prepare(W)
            ta = timeit(lambda: _hip.vq_assign(x, W, prepared=prep))
            tf = timeit(lambda: _hip.vq_forward(x, W, True, 1.0, prepared=prep))
            dead = float(_hip.vq_forward(x, W, True, 1.0, prepared=prep)[3])
            es = 2 if dt == torch.bfloat16 else 4
            byts = n * c * es * 2 + n * 8
            print(f"{str(dt)[6:]:9s} N{n} C{c} {usage:12s} dead {dead:5.1f}%  assign {ta:7.1f} us  gather+finalize {tf - ta:7.1f} us  ({byts / (tf - ta) / 1e6:6.2f} TB/s algorithmic)", flush=True)
