"""A/B timing of vqseg_conv2d_f dispatch options on the decoder's 3x3 layers (bf16, B=32): variants are interleaved
in one process and repeated, medians reported, so clock drift hits all variants alike.
usage: python tools/ab_conv.py "opt=val[,opt=val]" "opt=val" ..."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
DEFAULTS = {"conv3x3_patch_tile512": 2, "conv3x3_patch_tile512_min_workgroups": 512, "conv3x3_patch_min_workgroups": 256, "conv3x3_patch_wide_tile": 0, "conv3x3_patch_unroll": 1, "conv3x3_patch_xcd_pair": 0}
variants = [dict(DEFAULTS, **dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv)) for v in (sys.argv[1:] or [""])]
LAYERS = [("dec0.0", 2048, 0, 1024, 16), ("dec0.1", 1024, 0, 1024, 16), ("dec1.0", 1024, 1024, 512, 32), ("dec1.1", 512, 0, 512, 32),
          ("dec2.0", 512, 512, 256, 64), ("dec2.1", 256, 0, 256, 64), ("dec3.0", 256, 256, 128, 128), ("dec3.1", 128, 0, 128, 128),
          ("l3.conv2", 256, 0, 256, 32), ("l2.conv2", 128, 0, 128, 64)]
B = 32
st = torch.cuda.current_stream().cuda_stream
for name, c1, c2, cout, hw in LAYERS:
    cin = c1 + c2
    x = torch.randn(B, hw, hw, c1, device=dev).bfloat16()
    x2 = torch.randn(B, hw, hw, c2, device=dev).bfloat16() if c2 else None
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    hi = torch.empty(L.vqseg_conv_packed_elems(cout, cin, 3, 3, 0), dtype=torch.int16, device=dev)
    assert L.vqseg_conv_pack_weights_f32(w.data_ptr(), cout, cin, 3, 3, 0, hi.data_ptr(), None, st) == 0
    y = torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device=dev)
    stat = torch.empty(L.vqseg_conv_stat_slots(B * hw * hw, cout) * 2 * cout, device=dev)

    def run():
        rc = L.vqseg_conv2d_f(x.data_ptr(), x2.data_ptr() if c2 else None, c1, hi.data_ptr(), None, y.data_ptr(), stat.data_ptr(),
                              B, hw, hw, cin, cout, 3, 3, 1, 1, 0, 1, hw, hw, 0, st)
        assert rc == 0, L.vqseg_last_error()
    times = [[] for _ in variants]
    for rep in range(7):
        for vi, v in enumerate(variants):
            for k, val in v.items():
                assert L.vqseg_set_option(k.encode(), val) >= 0, k
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[vi].append(e0.elapsed_time(e1) / 10 * 1e3)
    gf = 2.0 * B * hw * hw * cout * cin * 9 / 1e9
    print(f"{name:9s} {gf:7.1f} GF " + " | ".join(f"{statistics.median(t):8.1f} us {gf / statistics.median(t) * 1e3:7.1f} TF/s" for t in times), flush=True)
