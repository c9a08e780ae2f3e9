"""Functional run of the other BASELINE.json configurations through CPSTrainer (a few steps, synthetic data):
   python tools/run_config.py <v1|v2> <size> <K> <batch> [fp32]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed

recipe, size, k, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
amp = None if (len(sys.argv) > 5 and sys.argv[5] == "fp32") else torch.bfloat16
dev = torch.device("cuda:0")
name = "vqreptunet1x1" if recipe == "v1" else "vqreptunet1x1v2"
model = {"name": name, "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                  "vq_cfg": {"num_embeddings": [0, 0, k, k, k], "distance": "euclidean", "kmeans_init": True},
                                  "margin": 1.5, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
tr = CPSTrainer(CPSConfig(model=model, recipe=recipe, total_iters=10, amp_dtype=amp), dev)
data = SyntheticCropWeed(size, batch, dev, seed=3)
(l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
for i in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    out = tr.step(l_in, l_tg, ul_in)
    torch.cuda.synchronize()
    print(f"step {i}: {time.time() - t0:.3f}s loss {out['loss'].item():.4f} miou {out['miou'].item():.3f}", flush=True)
print("ok", recipe, size, k, batch, "bf16" if amp else "fp32", f"{2 * batch / (time.time() - t0):.1f} img/s (last step)")
