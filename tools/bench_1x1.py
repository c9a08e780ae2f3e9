"""Forward timing of the encoder's 1x1 convolutions and other non-patch layers (bf16, B=32) through vqseg_conv2d_f:
reports time, TFLOP/s and the HBM bytes (x + y) per second, operands rotated through 4 buffer sets (HBM-cold)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
B = 32
# name, cin, cout, hw_in, k, stride, pad
LAYERS = [("l1.conv1", 256, 64, 128, 1, 1, 0), ("l1.conv3", 64, 256, 128, 1, 1, 0), ("l1.down", 64, 256, 128, 1, 1, 0),
          ("l2.conv1", 512, 128, 64, 1, 1, 0), ("l2.conv3", 128, 512, 64, 1, 1, 0), ("l2.down", 256, 512, 128, 1, 2, 0),
          ("l3.conv1", 1024, 256, 32, 1, 1, 0), ("l3.conv3", 256, 1024, 32, 1, 1, 0), ("l4.conv1", 2048, 512, 16, 1, 1, 0),
          ("l4.conv3", 512, 2048, 16, 1, 1, 0), ("l1.conv2", 64, 64, 128, 3, 1, 1), ("l4.conv2", 512, 512, 16, 3, 1, 1),
          ("l2.conv2s2", 128, 128, 128, 3, 2, 1), ("dec4.0", 192, 32, 256, 3, 1, 1), ("dec4.1", 32, 32, 256, 3, 1, 1),
          ("dec4.0 dgA", 32, 128, 256, 3, 1, 1), ("dec4.0 dgB", 32, 64, 256, 3, 1, 1), ("dec3.1", 128, 128, 128, 3, 1, 1)]
st = torch.cuda.current_stream().cuda_stream
SETS = 4
for name, cin, cout, hw, k, s, p in LAYERS:
    ho = (hw + 2 * p - k) // s + 1
    xs = [torch.randn(B, hw, hw, cin, device=dev).bfloat16() for _ in range(SETS)]
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    hi = torch.empty(L.vqseg_conv_packed_elems(cout, cin, k, k, 0), dtype=torch.int16, device=dev)
    assert L.vqseg_conv_pack_weights_f32(w.data_ptr(), cout, cin, k, k, 0, hi.data_ptr(), None, st) == 0
    ys = [torch.empty(B, ho, ho, cout, dtype=torch.bfloat16, device=dev) for _ in range(SETS)]
    stat = torch.empty(L.vqseg_conv_stat_slots(B * ho * ho, cout) * 2 * cout, device=dev)
    it = [0]

    def run():
        it[0] += 1
        x, y = xs[it[0] % SETS], ys[it[0] % SETS]
        rc = L.vqseg_conv2d_f(x.data_ptr(), None, cin, hi.data_ptr(), None, y.data_ptr(), stat.data_ptr(), B, hw, hw, cin, cout, k, k, s, p,
                              0, 1, ho, ho, 0, st)
        assert rc == 0, L.vqseg_last_error()
    ts = []
    for rep in range(5):
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8 * 1e3)
    us = statistics.median(ts)
    gf = 2.0 * B * ho * ho * cout * cin * k * k / 1e9
    gb = (B * hw * hw * cin + B * ho * ho * cout) * 2 / 1e9
    print(f"{name:11s} {cin:5d}->{cout:5d} @{hw:3d} k{k} s{s}  {us:8.1f} us  {gf / us * 1e3:7.1f} TF/s  {gb / us * 1e3:6.2f} TB/s (x+y)", flush=True)
