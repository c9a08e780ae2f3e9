"""CPU restatement of one CPS training iteration (TEST INFRASTRUCTURE / reported CPU baseline only).

v1 recipe: deprecated/train_with_test_pt_pseudo_entropy_reg.py:141-203 (the driver of `vqreptunet1x1`,
SURVEY 3.1) on top of oracle/torch_ref.py's functional model: two models, eval passes for pseudo
labels, four training forwards, entropy-percentile pseudo labels, Dice criterion, commitment and
prototype terms, one backward, two Adam steps.  Plain torch fp32 CPU ops + autograd.
"""
from __future__ import annotations

import numpy as np
import torch

from . import torch_ref as R


def _trainable(p):
    return [v for k, v in p.items() if v.is_floating_point() and "running" not in k and "codebook" not in k
            and "num_batches" not in k and not k.startswith("prototype_loss.")]


def regularized_pseudo_label(raw, percent):
    prob = torch.softmax(raw, dim=1)
    label = torch.argmax(prob, dim=1)
    entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
    thresh = np.percentile(entropy.detach().cpu().numpy().flatten(), percent)
    label[entropy.ge(thresh).bool()] = 255
    return label


class CPSReference:
    def __init__(self, state_dicts, num_embeddings=(0, 0, 512, 512, 512), lr=1e-4, margin=0.0, scale=1.0,
                 drop_percent=20.0, commitment_w=1.0, proto_w=0.01, cps_w=1.0):
        self.p = [{k: v.clone() for k, v in sd.items()} for sd in state_dicts]
        for p in self.p:
            for v in _trainable(p):
                v.requires_grad_(True)
        self.opt = [torch.optim.Adam(_trainable(p), lr=lr, betas=(0.9, 0.999)) for p in self.p]
        self.ks, self.margin, self.scale = num_embeddings, margin, scale
        self.drop_percent, self.cw, self.pw, self.cps_w = drop_percent, commitment_w, proto_w, cps_w

    def _fwd(self, p, x, training, gt=None, percent=None):
        return R.vq_unet_forward(p, x, training, self.ks, gt=gt, version=1, percent=percent, margin=self.margin,
                                 scale=self.scale)

    def step(self, l_input, l_target, ul_input, epoch_frac=0.0):
        p1, p2 = self.p
        for o in self.opt:
            o.zero_grad()
        with torch.no_grad():
            pseudo_1 = torch.argmax(self._fwd(p1, ul_input, False)[0], dim=1)
            pseudo_2 = torch.argmax(self._fwd(p2, ul_input, False)[0], dim=1)
        percent = 100 - self.drop_percent * (1 - epoch_frac)
        ps1, c_l1, _, q_l1, _ = self._fwd(p1, l_input, True, l_target, percent)
        ps2, c_l2, _, q_l2, _ = self._fwd(p2, l_input, True, l_target, percent)
        pu1, c_u1, _, q_u1, _ = self._fwd(p1, ul_input, True, pseudo_2, percent)
        pu2, c_u2, _, q_u2, _ = self._fwd(p2, ul_input, True, pseudo_1, percent)
        pred_1, pred_2 = torch.cat([ps1, pu1]), torch.cat([ps2, pu2])
        pl1, pl2 = regularized_pseudo_label(pred_1, percent), regularized_pseudo_label(pred_2, percent)
        cps = R.dice_loss(pred_1, pl2) + R.dice_loss(pred_2, pl1)
        sup = R.dice_loss(ps1, l_target) + R.dice_loss(ps2, l_target)
        commitment = (c_l1 + c_l2 + c_u1 + c_u2) * self.cw
        proto = (q_l1 + q_l2 + q_u1 + q_u2) * self.pw
        loss = sup + self.cps_w * cps + commitment.sum() + proto.float()
        loss.backward()
        for o in self.opt:
            o.step()
        return float(loss.detach())
