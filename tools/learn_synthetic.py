import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
dev = torch.device('cuda:0')
model = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
         "vq_cfg": {"num_embeddings": [0, 0, 64, 64, 64], "distance": "euclidean", "kmeans_init": True},
         "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
torch.manual_seed(0)
cfg = CPSConfig(model=model, recipe="v1", total_iters=200, amp_dtype=torch.bfloat16, learning_rate=1e-3)
tr = CPSTrainer(cfg, dev)
data = SyntheticCropWeed(128, 8, dev, seed=5)
t0 = time.time()
for i in range(120):
    (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
    out = tr.step(l_in, l_tg, ul_in, epoch_frac=i / 120)
    if i % 10 == 0 or i == 119:
        print(i, f"loss {out['loss'].item():.3f} sup {out['sup_loss_1'].item():.3f} miou {out['miou'].item():.3f}", flush=True)
print('time', time.time() - t0)
