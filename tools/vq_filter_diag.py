"""Candidate pairs per row the bf16 candidate filter hands to the exact re-score on the TRAINER's own features (bench configuration, smaller
batch): python tools/vq_filter_diag.py [batch] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vq_seg_amd import _hip
from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
tr = CPSTrainer(CPSConfig(model=bench.model_cfg(), recipe="v1", total_iters=100, amp_dtype=torch.bfloat16), dev)
data = SyntheticCropWeed(512, B, dev, seed=42)
for i in range(steps):
    (l_in, l_tg), ul = data.labelled(), data.unlabelled()
    _hip.FILTER_DIAG = []
    out = tr.step(l_in, l_tg, ul)
    torch.cuda.synchronize()
    line = []
    for n, c, k, cnt in _hip.FILTER_DIAG:
        line.append(f"N{n}xC{c}: " + ("-" if cnt is None else f"{int(cnt.sum()) / n:5.3f}"))
    print(f"step {i} loss {float(out['loss']):.4f} pairs per row: " + " | ".join(line), flush=True)
