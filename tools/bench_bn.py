"""BatchNorm kernels on the bench's layer shapes (bf16, HBM-cold operands by rotating buffer sets): apply (y -> out), backward (reduce + apply)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=12):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i + 1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot_a = tot_b = 0.0
for (M, C, cnt, res) in ((32 * 256 * 256, 64, 1, 0), (32 * 128 * 128, 64, 6, 0), (32 * 128 * 128, 256, 4, 1), (32 * 64 * 64, 128, 8, 0), (32 * 64 * 64, 512, 5, 1),
                         (32 * 32 * 32, 256, 12, 0), (32 * 32 * 32, 1024, 7, 1), (32 * 16 * 16, 512, 6, 0), (32 * 16 * 16, 2048, 4, 1), (32 * 16 * 16, 1024, 2, 0),
                         (32 * 32 * 32, 512, 2, 0), (32 * 64 * 64, 256, 2, 0), (32 * 128 * 128, 128, 2, 0), (32 * 256 * 256, 32, 2, 0)):
    SETS = max(2, min(8, int(1.2e9 / (M * C * 2 * 3))))
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
    ys = [torch.randn(M, C, device=dev).bfloat16() for _ in range(SETS)]
    gs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(SETS)]
    rs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(SETS)] if res else None
    outs = [torch.empty_like(t) for t in ys]
    mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev); gamma = torch.ones(C, device=dev)
    ws = torch.empty(L.vqseg_bn_backward_workspace_floats(M, C), device=dev)
    dg = torch.empty(2, C, device=dev); gy = torch.empty_like(ys[0]); gres = torch.empty_like(ys[0]) if res else None
    def apply(i):
        k = i % SETS
        assert L.vqseg_bn_apply_f(1, ys[k].data_ptr(), rs[k].data_ptr() if res else None, sc.data_ptr(), sh.data_ptr(), M, C, 1, outs[k].data_ptr(), st) == 0
    def bwd(i):
        k = i % SETS
        assert L.vqseg_bn_backward_f(1, gs[k].data_ptr(), outs[k].data_ptr() if res else None, ys[k].data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(),
                                     sc.data_ptr(), sh.data_ptr(), M, C, 1, 1, 0, ws.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(), gy.data_ptr(),
                                     gres.data_ptr() if res else None, None, st) == 0
    ta, tb = timeit(apply), timeit(bwd)
    ba = M * C * 2 * (3 if res else 2); bb = M * C * 2 * ((3 if res else 2) + (5 if res else 3))
    tot_a += ta * cnt; tot_b += tb * cnt
    print(f"M={M:8d} C={C:5d} x{cnt:2d} res={res}  apply {ta:7.1f} us {ba / ta / 1e6:5.2f} TB/s   backward {tb:7.1f} us {bb / tb / 1e6:5.2f} TB/s", flush=True)
print(f"one model pass (weighted by layer counts): apply {tot_a / 1e3:.2f} ms, backward {tot_b / 1e3:.2f} ms")
