"""How long the host needs to ENQUEUE one CPS step (no device sync) against how long the device needs to run it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tr = CPSTrainer(CPSConfig(model=bench.model_cfg(), recipe="v1", total_iters=20, amp_dtype=torch.bfloat16), dev)
data = SyntheticCropWeed(512, B, dev, seed=42)
(l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
for _ in range(3):
    tr.step(l_in, l_tg, ul_in)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    tr.step(l_in, l_tg, ul_in)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e3 * (t1 - t0):.1f} ms, device done after {1e3 * (t2 - t0):.1f} ms", flush=True)
