"""conv_pack_all_kernel per layer shape (us, GB/s of fp32 read + bf16 images written)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
L = _hip.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
for cout, cin, k in ((1024, 2048, 3), (512, 2048, 3), (512, 512, 3), (128, 128, 3), (64, 64, 3), (2048, 512, 1), (256, 64, 1), (64, 256, 1)):
    w = torch.randn(cout, cin, k, k, device=dev)
    cin_p, cout_p = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
    f = torch.empty(cout * k * k * cin_p, dtype=torch.int16, device=dev); t = torch.empty(cin * k * k * cout_p, dtype=torch.int16, device=dev)
    s3 = torch.empty(cout * k * k * 3 * cin, dtype=torch.int16, device=dev)
    run = lambda: L.vqseg_conv_pack_all_f32(w.data_ptr(), cout, cin, k, cin, f.data_ptr(), t.data_ptr(), s3.data_ptr(), st)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = w.numel() * (4 + 2 + 2 + 6) / 1e9
    print(f"{cout:5d} x {cin:5d} k{k}: {us:8.1f} us  {gb / us * 1e3:6.2f} TB/s")
