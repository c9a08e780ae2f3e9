"""Per-layer-shape table of the convolution launches of one CPS step (single stream, in-stream HIP events; bench.py's roofline_conv
leg with shapes):   python tools/conv_layers.py [batch]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vq_seg_amd import _hip
from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
tr = CPSTrainer(CPSConfig(model=bench.model_cfg(), recipe="v1", total_iters=10, amp_dtype=torch.bfloat16), dev)
data = SyntheticCropWeed(512, B, dev, seed=42)
(l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
tr._two_streams = False
for _ in range(2): tr.step(l_in, l_tg, ul_in)
torch.cuda.synchronize()
_hip.conv_profile_begin(4096)
tr.step(l_in, l_tg, ul_in)
torch.cuda.synchronize()
recs = _hip.conv_profile_collect(4096, with_shape=True)
agg = collections.OrderedDict()
for fl, kd, ms, sh in recs:
    a = agg.setdefault((kd, sh), [0, 0.0, 0.0])
    a[0] += 1; a[1] += fl; a[2] += ms
tot = sum(v[2] for v in agg.values())
print(f"{len(recs)} launches, {tot:.1f} ms")
for (kd, sh), (n, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"k{kd // 100} { {0: 'bf16', 1: 'prec', 2: 's3', 50: 'wgrad', 51: 'wgrad-prec'}[kd % 100]:5s} px {sh[0]:5d}k cin {sh[1]:5d} cout {sh[2]:5d} s{sh[3] // 10} up{sh[3] % 10}  x{n:3d}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
