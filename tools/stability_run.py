"""Long run at the benchmark's size with fresh synthetic data every step: allocator flat?  every term finite?  loss moving?
   python tools/stability_run.py [steps=200]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
tr = CPSTrainer(CPSConfig(model=bench.model_cfg(), recipe="v1", total_iters=steps, amp_dtype=torch.bfloat16), dev)
data = SyntheticCropWeed(512, 32, dev, seed=42)
t0 = time.time()
for i in range(steps):
    (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
    out = tr.step(l_in, l_tg, ul_in, epoch_frac=i / steps)
    if i % 20 == 0 or i == steps - 1:
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in out.items()}
        assert all(v == v and abs(v) < 1e30 for v in vals.values()), vals
        print(f"step {i:4d}  {time.time() - t0:6.1f}s  loss {vals['loss']:.4f} sup {vals['sup_loss_1']:.4f} cps {vals['cps_loss']:.4f} miou {vals['miou']:.3f}  "
              f"allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  peak {torch.cuda.max_memory_allocated() / 2**30:.1f}  reserved {torch.cuda.memory_reserved() / 2**30:.1f}", flush=True)
