import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
from tools.bench_vq import timeit
dev = torch.device("cuda:0")
for (n, c, k) in [(131072, 128, 512), (131072, 256, 512), (131072, 512, 512), (131072, 1024, 512), (131072, 2048, 512), (65536, 512, 512), (262144, 512, 512), (524288, 512, 512), (131072, 512, 256), (131072, 512, 1024), (131072, 512, 2048)]:
    x = torch.relu(torch.randn(n, c, device=dev)); W = torch.relu(torch.randn(k, c, device=dev))
    prep = _hip.vq_prepare(W)
    t = timeit(lambda: _hip.vq_assign(x, W, prepared=prep), iters=10)
    fl = 2.0 * n * c * k
    print(f"N={n:7d} C={c:5d} K={k:5d}  {t*1e3:9.1f} us  {fl/t/1e9:7.1f} TF/s  wgs={(n//128)*max(k//256,1)}")
