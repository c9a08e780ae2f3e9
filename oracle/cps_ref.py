"""CPU restatement of one CPS training iteration (TEST INFRASTRUCTURE / reported CPU baseline only).

On top of oracle/torch_ref.py's functional model (plain torch fp32 CPU ops + autograd):
  v1 recipe: deprecated/train_with_test_pt_pseudo_entropy_reg.py:141-203 (the driver of `vqreptunet1x1`): eval
             passes -> argmax pseudo labels, four training forwards with `percent`, entropy-percentile CPS pseudo labels
             (:30-39), criterion from the config (Dice), commitment and prototype terms, one backward, two Adam steps;
  v2 recipe: train_vqreptunet1x1v2.py:137-211: eval passes -> pseudo SCORES, four training forwards with `th`, score-mask
             CPS pseudo labels (:43-46), 0.5 CE + Dice.  (The reference's own v2 backward raises on fp32; this
             restatement's out-of-place prototype loss is differentiable.)

Pinned by tests/golden/cps_iter_v{1,2}.npz, which oracle/make_golden.py captured by driving the reference's OWN modules
through the loop body restated in tests/cps_loop.py (tests/test_oracle_golden.py::test_cps_iterations).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import torch_ref as R


def _trainable(p, version=1):
    """model.parameters() minus what never receives a gradient: running statistics are buffers; the codebooks get none
    (vq_img.py:236-239) and v1's prototypes enter through `.data` (prototype.py:556) -- Adam skips grad-None parameters."""
    keep = []
    for k, v in p.items():
        if not v.is_floating_point() or "running" in k or "num_batches" in k or "codebook" in k:
            continue
        if version == 1 and k.startswith("prototype_loss."):
            continue
        keep.append(v)
    return keep


def regularized_pseudo_label(raw, percent):
    """make_regularized_pseudo_label, deprecated/train_with_test_pt_pseudo_entropy_reg.py:30-39."""
    prob = torch.softmax(raw, dim=1)
    label = torch.argmax(prob, dim=1)
    entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
    thresh = np.percentile(entropy.detach().cpu().numpy().flatten(), percent)
    label[entropy.ge(thresh).bool()] = 255
    return label


class CPSReference:
    def __init__(self, state_dicts, num_embeddings=(0, 0, 512, 512, 512), lr=1e-4, min_lr=1e-7, total_iters=1000, margin=0.0,
                 scale=1.0, drop_percent=20.0, commitment_w=1.0, proto_w=0.01, cps_w=1.0, version=1, th=0.7):
        self.version = version
        self.p = [{k: v.clone() for k, v in sd.items()} for sd in state_dicts]
        for p in self.p:
            for v in _trainable(p, version):
                v.requires_grad_(True)
        self.opt = [torch.optim.Adam(_trainable(p, version), lr=lr, betas=(0.9, 0.999)) for p in self.p]
        self.ks, self.margin, self.scale, self.th = num_embeddings, margin, scale, th
        self.drop_percent, self.cw, self.pw, self.cps_w = drop_percent, commitment_w, proto_w, cps_w
        self.lr, self.min_lr, self.total_iters, self.it = lr, min_lr, total_iters, 0

    def _fwd(self, p, x, training, gt=None, percent=None):
        return R.vq_unet_forward(p, x, training, self.ks, gt=gt, version=self.version, percent=percent, th=self.th,
                                 margin=self.margin, scale=self.scale)

    def _ce_dice(self, pred, target):
        return 0.5 * F.cross_entropy(pred, target, ignore_index=255) + R.dice_loss(pred, target)

    def step(self, l_input, l_target, ul_input, epoch_frac=0.0, backward=True) -> Dict[str, object]:
        p1, p2 = self.p
        for o in self.opt:
            o.zero_grad()
        with torch.no_grad():
            score_1, score_2 = self._fwd(p1, ul_input, False)[0], self._fwd(p2, ul_input, False)[0]
        percent = 100 - self.drop_percent * (1 - epoch_frac)
        if self.version == 1:
            gt_1, gt_2 = torch.argmax(score_2, dim=1), torch.argmax(score_1, dim=1)
        else:
            gt_1, gt_2 = score_2, score_1
        ps1, c_l1, _, q_l1, _ = self._fwd(p1, l_input, True, l_target, percent)
        ps2, c_l2, _, q_l2, _ = self._fwd(p2, l_input, True, l_target, percent)
        pu1, c_u1, _, q_u1, _ = self._fwd(p1, ul_input, True, gt_1, percent)
        pu2, c_u2, _, q_u2, _ = self._fwd(p2, ul_input, True, gt_2, percent)
        pred_1, pred_2 = torch.cat([ps1, pu1]), torch.cat([ps2, pu2])
        if self.version == 1:
            mask_1, mask_2 = regularized_pseudo_label(pred_1, percent), regularized_pseudo_label(pred_2, percent)
            cps = R.dice_loss(pred_1, mask_2) + R.dice_loss(pred_2, mask_1)
            sup_1, sup_2 = R.dice_loss(ps1, l_target), R.dice_loss(ps2, l_target)
        else:
            mask_1 = R.score_mask(pred_1, torch.argmax(pred_1, dim=1).long(), self.th)
            mask_2 = R.score_mask(pred_2, torch.argmax(pred_2, dim=1).long(), self.th)
            cps = self._ce_dice(pred_1, mask_2) + self._ce_dice(pred_2, mask_1)
            sup_1, sup_2 = self._ce_dice(ps1, l_target), self._ce_dice(ps2, l_target)
        commitment = (c_l1 + c_l2 + c_u1 + c_u2) * self.cw
        proto = (q_l1 + q_l2 + q_u1 + q_u2) * self.pw
        lr = R.cosine_lr(self.it, self.lr, self.min_lr, self.total_iters)
        for o in self.opt:
            o.param_groups[0]["lr"] = lr
        loss = sup_1 + sup_2 + self.cps_w * cps + commitment.sum() + proto.float()
        if backward:
            loss.backward()
            for o in self.opt:
                o.step()
        self.it += 1
        conf = R.confusion_matrix(ps1.detach().numpy(), l_target.numpy())
        return {"loss": float(loss.detach()), "sup_loss_1": float(sup_1.detach()), "sup_loss_2": float(sup_2.detach()),
                "cps_loss": float(cps.detach()), "commitment_loss": float(commitment.detach().sum()),
                "prototype_loss": float(proto.detach()), "lr": lr, "step_miou": R.miou(conf)[0], "mask_1": mask_1, "mask_2": mask_2,
                "score_1": score_1, "pred_sup_1": ps1.detach(), "pred_ul_2": pu2.detach()}

    def evaluate(self, images, labels) -> float:
        """test() of the trainers (train_vqreptunet1x1v2.py:28-41): mean of the per-image mIoU of model_1."""
        total = 0.0
        with torch.no_grad():
            for i in range(images.shape[0]):
                pred = self._fwd(self.p[0], images[i:i + 1], False)[0]
                total += R.miou(R.confusion_matrix(pred.numpy(), labels[i:i + 1].numpy()))[0]
        return total / images.shape[0]
