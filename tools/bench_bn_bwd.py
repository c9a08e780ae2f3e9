"""Micro-benchmark of vqseg_bn_backward_f (bf16) with and without the `out` tensor (mask recomputed from y)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for (M, C) in ((32 * 128 * 128, 64), (32 * 128 * 128, 256), (32 * 64 * 64, 256), (32 * 32 * 32, 1024), (32 * 256 * 256, 32)):
    SETS = 6                                               # rotate buffer sets so that the inputs come from HBM, not the MALL
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
    ys = [torch.randn(M, C, device=dev).bfloat16() for _ in range(SETS)]
    gs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(SETS)]
    outs = [torch.relu(t.float() * sc + sh).bfloat16() for t in ys]
    y, g, out = ys[0], gs[0], outs[0]
    mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev); gamma = torch.ones(C, device=dev)
    ws = torch.empty(L.vqseg_bn_backward_workspace_floats(M, C), device=dev)
    dg = torch.empty(2, C, device=dev); gy = torch.empty_like(y)
    for use_out in (True, False):
        it = [0]
        def run():
            it[0] += 1
            y, g, out = ys[it[0] % SETS], gs[it[0] % SETS], outs[it[0] % SETS]
            rc = L.vqseg_bn_backward_f(1, g.data_ptr(), out.data_ptr() if use_out else None, y.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                       gamma.data_ptr(), sc.data_ptr(), sh.data_ptr(), M, C, 1, 1, 0, ws.data_ptr(), dg[0].data_ptr(),
                                       dg[1].data_ptr(), gy.data_ptr(), None, None, st)
            assert rc == 0, L.vqseg_last_error()
        it[0] = -1
        run()
        ref = gy.clone() if use_out else ref
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        it[0] = 0
        for _ in range(18):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 18 * 1e3
        gb = M * C * 2 * ((3 if use_out else 2) + (4 if use_out else 3)) / 1e9
        print(f"M={M} C={C} out={'given' if use_out else 'none '} {us:8.1f} us  {gb / us * 1e3:6.2f} TB/s  same={torch.equal(gy, ref)}", flush=True)
