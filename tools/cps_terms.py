"""Print every term of two CPSTrainer steps next to the reference fixture (debug aid for tests/test_cps_parity_gpu.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import cps_loop, golden_io
from tests.test_cps_parity_gpu import _trainer

version = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fx = golden_io.load(f"cps_iter_v{version}")
tr, dev = _trainer(version, False)
for i, (l_in, l_tg, ul_in) in enumerate(cps_loop.batches(2)):
    out = tr.step(l_in.to(dev), l_tg.to(dev), ul_in.to(dev), epoch_frac=0.0)
    for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss"):
        print(i, key, float(out[key]), float(fx[f"it{i}/{key}"]))
    for key in ("mask_1", "mask_2"):
        print(i, key, int((tr.aux[key].cpu().to(torch.uint8) != fx[f"it{i}/{key}"]).sum()))
    for key in ("score_1", "pred_sup_1", "pred_ul_2"):
        a, b = tr.aux[key].double().cpu(), fx[f"it{i}/{key}"].double()
        print(i, key, ((a - b).abs().max() / b.abs().max()).item())
