"""Evaluation loop (reference: test_detailviz.py:87-163 `test_loop`, without its wandb / visualisation side).

Eval-mode model (on the HIP path BatchNorm + residual + ReLU ride in the convolution epilogues, the VQ layers are pure
gathers), logits resized bilinearly to the native mask size, metrics per batch exactly as `Measurement.measure` defines
them (`measurement.py:7-91`: per-image confusion matrices, IoU with the +1e-8 denominator, batch means) and averaged
over the batches.  Only the (N, C, C) confusion counts and the per-image hit counts leave the device.
"""
from typing import Dict, Iterable, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import nnf
from .measurement import Measurement, confusion_matrix_device


@torch.no_grad()
def test_loop(model: torch.nn.Module, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], num_classes: int,
              device=None, amp_dtype=None) -> Dict[str, object]:
    """batches yields (images (N, 3, H, W) float in [0, 1], labels (N, Hm, Wm) int class ids)."""
    was_training = model.training
    model.eval()
    meas = Measurement(num_classes)
    sums = dict(test_acc=0.0, test_miou=0.0, test_precision=0.0, test_recall=0.0, test_f1score=0.0)
    iou_per_class = np.zeros(num_classes, dtype=np.float64)
    n_batches = 0
    for img, target in batches:
        if device is not None:
            img, target = img.to(device), target.to(device)
        if amp_dtype is not None and img.is_cuda:
            with torch.autocast("cuda", dtype=amp_dtype):
                pred = model(img)
        else:
            pred = model(img)
        pred = pred[0] if isinstance(pred, tuple) else pred
        pred = pred.float()
        # F.interpolate(pred, size, mode="bilinear") of test_detailviz.py:118 (align_corners=False): the HIP resize kernel on the device
        pred = nnf.upsample_bilinear(pred, size=tuple(target.shape[-2:]), align_corners=False) if pred.is_cuda else \
            F.interpolate(pred, target.shape[-2:], mode="bilinear")
        conf = confusion_matrix_device(pred, target, num_classes)                  # (N, C, C), rows = ground truth
        hits = (pred.argmax(dim=1) == target).flatten(1).double().mean(dim=1)       # Measurement.accuracy per image
        conf_np = conf.cpu().numpy()
        miou, ious = meas.miou(conf_np)
        precision, _ = meas.precision(conf_np)
        recall, _ = meas.recall(conf_np)
        sums["test_acc"] += float(hits.mean())
        sums["test_miou"] += float(miou)
        sums["test_precision"] += float(precision)
        sums["test_recall"] += float(recall)
        sums["test_f1score"] += float(meas.f1score(recall, precision))
        iou_per_class += np.array(ious)
        n_batches += 1
    model.train(was_training)
    if n_batches == 0:
        raise ValueError("test_loop: no batches")
    out = {k: v / n_batches for k, v in sums.items()}
    out["test_ious"] = np.round(iou_per_class / n_batches, 5).tolist()
    return out
