"""Reliable prototype losses (reference: models/modules/prototype.py:500-613 and :778-888).

Small (B*H/2*W/2 x 32) x (3 x 32) math kept as device-side PyTorch tensor ops; the
k-means prototype initialisation reuses the HIP k-means of the VQ layer.  Differences from the
reference, on purpose:
  * v1 takes the entropy-percentile threshold on the device (nnf.percentile: exact radix select of the two
    order statistics, np.percentile's linear interpolation) instead of numpy on the host (removes a host sync, q11);
  * v2 applies the margin out of place, so backward works in fp32 (the reference's in-place form
    raises, q10); forward values are identical.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from ... import nnf
from ...utils.seg_tools import onehot_1d
from ...vector_quantizer.vq_img import kmeans


def _rows(x):
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(b * h * w, c)


class _PrototypeBase(nn.Module):
    def __init__(self, num_classes, embedding_dim, scale, margin, init="kmeans", use_feature=False, easy_margin=True,
                 orthogonal_reg_weight=0):
        super().__init__()
        self.use_feature, self.num_classes = use_feature, num_classes
        self.scale, self.margin, self.init = scale, margin, init
        self.embedding = nn.Embedding(num_embeddings=num_classes, embedding_dim=embedding_dim)
        self.orthogonal_reg_weight = orthogonal_reg_weight
        self.initted = False
        if init == "uniform":
            self.embedding.weight.data.uniform_(-1 / num_classes, 1 / num_classes)
            self.initted = True
        elif init == "normal":
            self.embedding.weight.data.normal_()
            self.initted = True
        elif init != "kmeans":
            raise ValueError("init has to be in ['uniform', 'normal', 'kmeans']")
        self.easy_margin = easy_margin
        self.cos_m, self.sin_m = math.cos(margin), math.sin(margin)
        self.th = math.cos(math.pi - margin)
        self.mm = math.sin(math.pi - margin) * margin
        if use_feature or orthogonal_reg_weight > 0:
            raise NotImplementedError("use_feature / orthogonal_reg_weight are unused by the target configs")

    @torch.no_grad()
    def _kmeans_init(self, rows):
        if self.initted:
            return
        means, _ = kmeans(rows.detach().float().contiguous(), self.num_classes, 10)
        self.embedding.weight.data.copy_(means)
        self.initted = True

    def _phi(self, cosine):
        sine = torch.sqrt((1.0 - torch.pow(cosine, 2)).clamp(0, 1))
        phi = cosine * self.cos_m - sine * self.sin_m
        if self.easy_margin:
            return torch.where(cosine > 0, phi, cosine)
        return torch.where(cosine > self.th, phi, cosine - self.mm)


class ReliablePrototypeLoss(_PrototypeBase):
    """v1: entropy-percentile-filtered ArcFace-style prototype CE, computed in float64 (q11).
    Prototypes enter through `.data` => they receive no gradient (prototype.py:556)."""

    @torch.autocast("cuda", enabled=False)
    def forward(self, x, gt, percent, entropy):
        gt = gt.unsqueeze(1) if gt.dim() == 3 else gt
        if gt.shape != x.shape:
            gt = F.interpolate(gt.float(), x.shape[-2:], mode="nearest").long()
        if not self.initted and self.init == "kmeans" and self.training:
            self._kmeans_init(_rows(x.float()))
        proto = F.normalize(self.embedding.weight.data, p=2, dim=-1)
        if nnf.proto_loss_supported(x, self.num_classes):
            # HIP path: one fused pass forward, one backward (vqseg_proto_loss_*), bf16 or fp32 features as they are
            with torch.no_grad():
                keep = torch.le(entropy, nnf.percentile(entropy, percent))      # np.percentile by exact radix select
            return nnf.proto_loss(x, proto, _rows(gt), keep=keep, variant=1, scale=self.scale, margin=self.margin,
                                  easy_margin=self.easy_margin)
        x = x.float()
        rows = _rows(x)
        labels = _rows(gt)
        onehot = onehot_1d(labels, self.num_classes)
        rows = F.normalize(rows, p=2, dim=-1)
        cosine = F.linear(rows, proto)
        phi = self._phi(cosine)
        if self.margin != 0:
            cosine = (onehot * phi) + ((1.0 - onehot) * cosine)
        if self.scale != 1:
            cosine = self.scale * cosine
        with torch.no_grad():
            thresh = torch.quantile(entropy.detach().flatten().double(), percent / 100.0)   # == np.percentile (linear)
            keep = torch.le(entropy, thresh.to(entropy.dtype))
        positive = torch.exp(torch.sum(cosine * onehot, dim=-1))
        total = torch.sum(torch.exp(cosine), dim=-1)
        return -torch.mean(torch.log((positive / (total + 1e-7)) + 1e-7) * keep)


class ReliablePrototypeLossv2(_PrototypeBase):
    """v2: hard labels, or pseudo scores (B, C, H, W) with a confidence mask at threshold `th`."""

    @torch.autocast("cuda", enabled=False)
    def forward(self, x, gt, th):
        x_in = x
        x = x.float()
        conf = None
        if gt.dim() == 4:
            pred = gt
            if pred.shape[-2:] != x.shape[-2:]:
                pred = F.interpolate(pred.float(), x.shape[-2:], mode="bilinear")
            prob = torch.softmax(_rows(pred), dim=-1)
            conf = torch.where(prob.max(dim=1)[0] > th, 1, 0).to(pred.dtype)
            gt = torch.argmax(pred, dim=1)
        gt = gt.unsqueeze(1) if gt.dim() == 3 else gt
        if gt.shape[-2:] != x.shape[-2:]:
            gt = F.interpolate(gt.float(), x.shape[-2:], mode="nearest").long()
        labels = _rows(gt)[:, 0]
        if not self.initted and self.init == "kmeans" and self.training:
            self._kmeans_init(_rows(x.float()))
        self.embedding.weight.data = F.normalize(self.embedding.weight.data, p=2, dim=-1)   # prototype.py:844
        if nnf.proto_loss_supported(x_in, self.num_classes):
            return nnf.proto_loss(x_in, self.embedding.weight, labels, conf=conf, variant=2, scale=self.scale, margin=self.margin,
                                  easy_margin=self.easy_margin).float()
        rows = _rows(x)
        rows = F.normalize(rows, p=2, dim=-1)
        cosine = F.linear(rows, self.embedding.weight)
        phi = self._phi(cosine)
        hit = F.one_hot(labels, self.num_classes).bool()
        cosine = torch.where(hit, cosine * phi, cosine)                                    # :860, out of place
        cosine = self.scale * cosine
        positive = torch.exp(cosine.gather(1, labels[:, None])[:, 0])
        total = torch.sum(torch.exp(cosine), dim=-1)
        ll = torch.log((positive / (total + 1e-7)) + 1e-7)
        return -torch.mean(ll) if conf is None else -torch.mean(ll * conf)
