"""CPU oracle for the VQ-UNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement (plain torch fp32 CPU ops, parameters passed as a
state_dict-style mapping with the reference's key names) of the algorithm the
reference implements for the path BASELINE.json's north_star names.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / the reported CPU baseline.

Pinning: every function here is checked against golden vectors captured from the
reference's own modules imported in the build container
(`oracle/make_golden.py` -> `tests/golden/*.npz`, test: `tests/test_oracle_golden.py`).
The one exception is the ResNet-50 body (`resnet_encoder`), whose arithmetic lives
in torchvision 0.14.1 (absent here, un-vendored): that part is PARITY UNPINNED and
follows the published torchvision Bottleneck (v1.5) semantics.  `vq_ema_update` restates
an EXTENSION the reference does not have (opt-in EMA codebook update): PARITY UNPINNED too.

Citations are into /root/reference (read-only), `file:line`.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Mapping[str, Tensor]


# --------------------------------------------------------------------------------------
# Vector quantiser  (vector_quantizer/vq_img.py)
# --------------------------------------------------------------------------------------
def vq_lookup(rows: Tensor, codebook: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """EuclideanCodebook.forward, vq_img.py:160-177, for rows (..., C), codebook (K, C).

    Same op sequence as the reference: cdist -> argmin -> one_hot -> float -> matmul ->
    bincount -> dead-code percentage.
    """
    rows = rows.float()
    k = codebook.shape[0]
    dist = torch.cdist(rows, codebook, p=2)                       # :167
    idx = torch.argmin(dist, dim=-1)                              # :168  first minimum wins
    quant = torch.matmul(F.one_hot(idx, num_classes=k).float(), codebook)   # :169-170
    counts = torch.bincount(idx.reshape(-1), minlength=k)         # :173
    dead_pct = 100 * ((counts == 0).sum() / k)                    # :174-175
    return quant, idx, dead_pct


def vq_forward(x: Tensor, codebook: Tensor, training: bool, commitment_weight: float = 1.0):
    """VectorQuantizer.forward, vq_img.py:228-244.  x is (B, C, H, W)."""
    x = x.to(torch.float32)                                        # :229
    b, c, h, w = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(b, h * w, c)              # :232
    quant, idx, dead_pct = vq_lookup(rows, codebook)               # :233
    loss = torch.tensor([0.0], requires_grad=training, device=x.device)   # :234
    if training:
        quant = rows + (quant - rows).detach()                    # :236 straight-through
        if commitment_weight > 0:
            loss = loss + F.mse_loss(quant.detach(), rows) * commitment_weight   # :238-240
    quant = quant.reshape(b, h, w, c).permute(0, 3, 1, 2)          # :242
    idx = idx.reshape(b, h, w)                                     # :243
    return quant, idx, loss, dead_pct


def kmeans_lloyd(samples: Tensor, means0: Tensor, iters: int) -> Tuple[Tensor, Tensor]:
    """kmeans(), vq_img.py:29-63 (euclidean branch), GIVEN the initial means.

    The reference draws the initial means with torch.randperm (`sample_vectors`,
    :10-17), which no other device can reproduce; everything after that draw is restated
    here.  samples (N, C), means0 (K, C) -> (means (K, C), bins (K,) int64).
    """
    samples = samples.float()
    k, c = means0.shape
    means = means0.clone().float()
    bins = torch.zeros(k, dtype=torch.long)
    for _ in range(iters):
        score = -torch.cdist(samples[None], means[None], p=2)[0]             # :39
        buckets = torch.argmax(score, dim=-1)                               # :41
        bins = torch.zeros(k, dtype=torch.long).scatter_add_(0, buckets, torch.ones_like(buckets))  # :42
        empty = bins == 0                                                    # :44
        denom = bins.masked_fill(empty, 1)                                   # :45
        sums = torch.zeros(k, c, dtype=samples.dtype)
        sums.scatter_add_(0, buckets[:, None].expand(-1, c), samples)        # :51
        fresh = sums / denom[:, None]                                        # :53
        means = torch.where(empty[:, None], means, fresh)                    # :58-61
    return means, bins


def vq_ema_update(cluster_size: Tensor, embed_avg: Tensor, rows: Tensor, idx: Tensor, decay: float, eps: float):
    """EXTENSION, PARITY UNPINNED: the reference has no EMA update (vq_img.py:72,83 store `decay` and never read it), so
    there is nothing of the reference's to follow line by line.  This restates the published rule of the module the
    reference's quantiser descends from (vector-quantize-pytorch, EuclideanCodebook.forward: one-hot counts and sums,
    ema_inplace on both, laplace_smoothing of the counts) for the opt-in `ema_update=True` path.
    Returns (new cluster_size, new embed_avg, new codebook)."""
    k = cluster_size.shape[0]
    onehot = F.one_hot(idx.reshape(-1), k).to(rows.dtype)
    counts = onehot.sum(0)
    sums = onehot.t() @ rows
    cluster_size = cluster_size * decay + counts * (1 - decay)
    embed_avg = embed_avg * decay + sums * (1 - decay)
    total = cluster_size.sum()
    smoothed = (cluster_size + eps) / (total + k * eps) * total
    return cluster_size, embed_avg, embed_avg / smoothed[:, None]


def vq_backward(x: Tensor, q_ste: Tensor, g_quant: Tensor, g_loss: Tensor, commitment_weight: float):
    """Analytic gradient of VectorQuantizer.forward w.r.t. x:
    d/dx = g_quant (identity through the straight-through estimator)
         + g_loss * w * 2 (x - q_ste) / numel      (mse_loss(q.detach(), x), :239)
    All tensors in the (B, C, H, W) frame.
    """
    n = x.numel()
    return g_quant + g_loss * commitment_weight * 2.0 * (x - q_ste) / n


# --------------------------------------------------------------------------------------
# UNet decoder  (models/networks/unet/decoder.py)
# --------------------------------------------------------------------------------------
def _bn(x: Tensor, p: Dict[str, Tensor], prefix: str, training: bool, eps: float, momentum: float) -> Tensor:
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], training, momentum, eps)


def conv3x3_bn_relu(x: Tensor, p: Dict[str, Tensor], prefix: str, training: bool,
                    eps: float = 1e-5, momentum: float = 0.1) -> Tensor:  # `training` = the BN module's mode
    """conv_bn_relu, decoder.py:7-10: Conv3x3(pad 1, zeros, no bias) -> BatchNorm2d -> ReLU.
    Keys: `<prefix>.0.weight` (conv), `<prefix>.1.*` (bn).  Running stats in `p` are
    updated in place in training mode exactly like nn.BatchNorm2d does.
    """
    y = F.conv2d(x, p[prefix + ".0.weight"], None, stride=1, padding=1)
    return F.relu(_bn(y, p, prefix + ".1", training, eps, momentum))


def unet_decoder(p: Dict[str, Tensor], features: Sequence[Tensor], training: bool, prefix: str = "decoder",
                 eps: float = 1e-5, momentum: float = 0.1) -> Tensor:  # `training` = mode of the BatchNorm modules
    """UnetDecoder.forward, decoder.py:30-39.  `features` shallow -> deep (5 maps)."""
    feats = list(features)[::-1]                                   # :31
    n_blocks = len(feats)
    cat = feats[0]
    out = None
    for i in range(n_blocks):
        out = conv3x3_bn_relu(cat, p, f"{prefix}.blocks.{i}.0", training, eps, momentum)
        out = conv3x3_bn_relu(out, p, f"{prefix}.blocks.{i}.1", training, eps, momentum)
        if i + 1 < n_blocks:
            up = F.interpolate(out, feats[i + 1].shape[-2:], mode="bilinear")   # :35 align_corners=False
            cat = torch.cat((up, feats[i + 1]), dim=1)                            # :35-37 upsampled first
    return out


# --------------------------------------------------------------------------------------
# ResNet encoder  (models/encoders/resnet.py:117-190 on torchvision 0.14 ResNet)  -- PARITY UNPINNED body
# --------------------------------------------------------------------------------------
RESNET_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}


def _conv(x: Tensor, w: Tensor, stride: int, pad: int, reflect: bool) -> Tensor:
    if pad and reflect:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")        # nn.Conv2d(padding_mode='reflect')
        pad = 0
    return F.conv2d(x, w, None, stride=stride, padding=pad)


def _bottleneck(x: Tensor, p: Dict[str, Tensor], prefix: str, stride: int, reflect: bool,
                training: bool, eps: float, momentum: float) -> Tensor:
    """torchvision Bottleneck v1.5: 1x1 -> 3x3(stride) -> 1x1(x4), projection shortcut when
    the shape changes.  resnet.py:143-148 flips only the block's own convs to reflect; the
    downsample conv (inside a nested Sequential) stays zero-padded (it is 1x1, pad 0)."""
    idt = x
    y = F.relu(_bn(_conv(x, p[prefix + ".conv1.weight"], 1, 0, reflect), p, prefix + ".bn1", training, eps, momentum))
    y = F.relu(_bn(_conv(y, p[prefix + ".conv2.weight"], stride, 1, reflect), p, prefix + ".bn2", training, eps, momentum))
    y = _bn(_conv(y, p[prefix + ".conv3.weight"], 1, 0, reflect), p, prefix + ".bn3", training, eps, momentum)
    if (prefix + ".downsample.0.weight") in p:
        idt = _bn(_conv(x, p[prefix + ".downsample.0.weight"], stride, 0, False), p,
                  prefix + ".downsample.1", training, eps, momentum)
    return F.relu(y + idt)


def resnet_encoder(p: Dict[str, Tensor], x: Tensor, training: bool, prefix: str = "encoder",
                   arch: str = "resnet50", reflect: bool = True, eps: float = 1e-5,
                   momentum: float = 0.1) -> List[Tensor]:
    """ResNetEncoder.forward, resnet.py:173-181 -> [x, stem, layer1..layer4] (6 maps)."""
    feats = [x]
    y = F.relu(_bn(_conv(x, p[f"{prefix}.conv1.weight"], 2, 3, reflect), p, f"{prefix}.bn1", training, eps, momentum))
    feats.append(y)                                                # :166
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)        # :167
    for li, nblk in enumerate(RESNET_LAYERS[arch], start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            y = _bottleneck(y, p, f"{prefix}.layer{li}.{bi}", stride, reflect, training, eps, momentum)
        feats.append(y)
    return feats


# --------------------------------------------------------------------------------------
# Prototype losses  (models/modules/prototype.py)
# --------------------------------------------------------------------------------------
def _margin_terms(cosine: Tensor, margin: float, easy_margin: bool = True) -> Tensor:
    cos_m, sin_m = math.cos(margin), math.sin(margin)
    sine = torch.sqrt((1.0 - cosine.pow(2)).clamp(0, 1))
    phi = cosine * cos_m - sine * sin_m
    if easy_margin:
        return torch.where(cosine > 0, phi, cosine)
    th, mm = math.cos(math.pi - margin), math.sin(math.pi - margin) * margin
    return torch.where(cosine > th, phi, cosine - mm)


def prototype_loss_v1(feat: Tensor, gt: Tensor, prototypes: Tensor, percent: float, entropy: Tensor,
                      margin: float, scale: float) -> Tensor:
    """ReliablePrototypeLoss.forward, prototype.py:531-598 (later definition wins),
    with the prototypes already initialised.  Result is float64 (onehot_1d is f64,
    utils/seg_tools.py:32).  The prototypes enter through `.data` (:556) => no gradient."""
    gt = gt.unsqueeze(1) if gt.dim() == 3 else gt
    if gt.shape != feat.shape:
        gt = F.interpolate(gt.float(), feat.shape[-2:], mode="nearest").long()      # :533-534
    c = feat.shape[1]
    n_cls = prototypes.shape[0]
    rows = feat.permute(0, 2, 3, 1).reshape(-1, c)
    labels = gt.permute(0, 2, 3, 1).reshape(-1, 1)
    onehot = torch.zeros(rows.shape[0], n_cls, dtype=torch.float64, device=rows.device).scatter_(1, labels, 1.0) + 1e-6   # seg_tools.py:23-34
    proto = F.normalize(prototypes.detach(), p=2, dim=-1)                          # :556
    rows = F.normalize(rows, p=2, dim=-1)                                           # :557
    cosine = F.linear(rows, proto)                                                  # :562
    phi = _margin_terms(cosine, margin)
    if margin != 0:
        cosine = (onehot * phi) + ((1.0 - onehot) * cosine)                         # :573
    if scale != 1:
        cosine = scale * cosine                                                     # :577
    thresh = float(np.percentile(entropy.detach().cpu().numpy().flatten(), percent))   # :582
    keep = torch.le(entropy, thresh)                                                # :585
    positive = torch.exp(torch.sum(cosine * onehot, dim=-1))                        # :591
    total = torch.sum(torch.exp(cosine), dim=-1)                                    # :592
    return -torch.mean(torch.log((positive / (total + 1e-7)) + 1e-7) * keep)        # :593


def prototype_loss_v2(feat: Tensor, gt: Tensor, prototypes: Tensor, th: Optional[float],
                      margin: float, scale: float) -> Tuple[Tensor, Tensor]:
    """ReliablePrototypeLossv2.forward, prototype.py:809-872, prototypes already initialised.
    Returns (loss, l2-normalised prototypes) -- the reference overwrites
    `embedding.weight.data` with the normalised value (:844).

    The reference's in-place column update (:860) makes its backward raise on fp32
    (SURVEY q10); the forward VALUE is what is pinned.  This restatement performs the
    same arithmetic out of place: cosine[gt] <- cosine[gt] * phi[gt].
    """
    conf = None
    if gt.dim() == 4:                                                               # :811-820 pseudo scores
        pred = gt
        if pred.shape[-2:] != feat.shape[-2:]:
            pred = F.interpolate(pred.float(), feat.shape[-2:], mode="bilinear")
        prob = torch.softmax(pred.permute(0, 2, 3, 1).reshape(-1, pred.shape[1]), dim=-1)
        conf = torch.where(prob.max(dim=1)[0] > th, 1, 0).to(pred.dtype)
        gt = torch.argmax(pred, dim=1)
    gt = gt.unsqueeze(1) if gt.dim() == 3 else gt
    if gt.shape[-2:] != feat.shape[-2:]:
        gt = F.interpolate(gt.float(), feat.shape[-2:], mode="nearest").long()      # :822-823
    c = feat.shape[1]
    rows = F.normalize(feat.permute(0, 2, 3, 1).reshape(-1, c), p=2, dim=-1)        # :845
    labels = gt.permute(0, 2, 3, 1).reshape(-1)
    proto = F.normalize(prototypes, p=2, dim=-1)                                    # :844
    cosine = F.linear(rows, proto)                                                  # :849
    phi = _margin_terms(cosine, margin)
    hit = F.one_hot(labels, proto.shape[0]).bool()
    cosine = torch.where(hit, cosine * phi, cosine)                                 # :860 (out of place)
    cosine = scale * cosine                                                         # :863
    positive = torch.exp(cosine.gather(1, labels[:, None])[:, 0])                   # :864
    total = torch.sum(torch.exp(cosine), dim=-1)                                    # :867
    ll = torch.log((positive / (total + 1e-7)) + 1e-7)
    loss = -torch.mean(ll) if conf is None else -torch.mean(ll * conf)              # :868
    return loss, proto


# --------------------------------------------------------------------------------------
# Losses, metrics, schedule  (loss/dice_loss.py, measurement.py, utils/lr_schedulers.py)
# --------------------------------------------------------------------------------------
def dice_loss(pred: Tensor, target: Tensor, ignore_index: int = 255) -> Tensor:
    """dice_loss / dice_coefficient, loss/dice_loss.py:5-57, 3 classes hard-coded (:17),
    weight=None branch.  Ignored pixels: logits zeroed and target forced to class 0 (q15)."""
    b, c, _, _ = pred.shape
    logits = pred.reshape(b, c, -1)
    tgt = target.reshape(b, -1)
    keep = tgt != ignore_index
    logits = logits * keep[:, None, :]
    tgt = tgt * keep
    onehot = torch.eye(c)[tgt.long()].permute(0, 2, 1)
    prob = F.softmax(logits, dim=1)
    inter = torch.sum(prob * onehot, dim=2)
    sets = torch.sum(prob + onehot, dim=2)
    dice = (2 * inter / (sets + 1e-6)).mean(dim=0)
    return 1 - dice.mean()


def score_mask(pred: Tensor, pseudo: Tensor, th: float = 0.7) -> Tensor:
    """train_vqreptunet1x1v2.py:43-46."""
    top = torch.softmax(pred, dim=1).max(dim=1)[0]
    return torch.where(top > th, pseudo, 255)


def confusion_matrix(pred: np.ndarray, target: np.ndarray, num_classes: int = 3) -> np.ndarray:
    """Measurement._make_confusion_matrix, measurement.py:12-31 -> (N, C, C) counts."""
    n = pred.shape[0]
    lab = pred.argmax(axis=1).reshape(n, -1)
    cat = num_classes * target.reshape(n, -1) + lab
    out = np.stack([np.bincount(row, minlength=num_classes ** 2) for row in cat])
    return out.reshape(n, num_classes, num_classes)


def miou(conf: np.ndarray) -> Tuple[float, List[float]]:
    """Measurement.miou, measurement.py:53-62."""
    col, row = conf.sum(-2), conf.sum(-1)
    ious = [float(np.mean(conf[:, i, i] / (col[:, i] + row[:, i] - conf[:, i, i] + 1e-8)))
            for i in range(conf.shape[-1])]
    return float(np.mean(np.array(ious))), ious


def cosine_lr(it: int, start_lr: float, min_lr: float, total_iters: int, warmup_steps: int = 0) -> float:
    """CosineAnnealingLR.get_lr, utils/lr_schedulers.py:110-112."""
    return min_lr + 0.5 * (start_lr - min_lr) * (1 + math.cos(math.pi * it / (float(total_iters) - warmup_steps)))


# --------------------------------------------------------------------------------------
# Whole models  (models/networks/modified_vqunet/net.py:1141-1222, :184-260; unet/net.py:806-838)
# --------------------------------------------------------------------------------------
def vq_unet_forward(p: Dict[str, Tensor], x: Tensor, training: bool, num_embeddings: Sequence[int],
                    gt: Optional[Tensor] = None, version: int = 1, percent: Optional[float] = None,
                    th: Optional[float] = None, margin: float = 0.0, scale: float = 1.0,
                    commitment_weight: float = 1.0, eps: float = 1e-5, momentum: float = 0.1,
                    features: Optional[Sequence[Tensor]] = None, bn_training: Optional[bool] = None):
    """VQRePTUnet1x1.forward (net.py:1174-1209) / VQRePTUnet1x1v2.forward (:217-247).

    Codebooks and prototypes are taken as already initialised (k-means init is RNG
    dependent and tested separately).  `features` short-circuits the encoder (pins
    everything downstream of it).  Returns (logits, commitment (1,), dead_pct (n_vq,),
    prototype_loss | None, aux dict with per-level indices and decoder output).
    """
    bn_tr = training if bn_training is None else bn_training      # BatchNorm modules may be frozen (.eval()) in a training model
    feats = list(features) if features is not None else resnet_encoder(p, x, bn_tr, eps=eps, momentum=momentum)[1:]
    loss = torch.zeros(1, device=x.device)
    usage, indices = [], []
    for i, k in enumerate(num_embeddings):
        if k == 0:
            continue                                                     # Identity, vector_quantizer/__init__.py:27-32
        q, idx, closs, dead = vq_forward(feats[i], p[f"codebook.{i}.codebook.embedding.weight"], training, commitment_weight)
        feats[i] = q
        loss = loss + closs                                              # :1188 / :230
        usage.append(dead.detach())
        indices.append(idx)
    loss = loss / len(feats)                                             # :1195 (divides by 5, q2)
    dec = unet_decoder(p, feats, bn_tr, eps=eps, momentum=momentum)
    logits = F.conv2d(dec, p["segmentation_head.weight"])               # 1x1, no bias
    proto = None
    if training:
        if version == 1:
            prob = torch.softmax(logits.detach().permute(0, 2, 3, 1).reshape(-1, logits.shape[1]), dim=1)
            entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)  # :1199-1202
            proto = prototype_loss_v1(dec, gt, p["prototype_loss.embedding.weight"], percent, entropy, margin, scale)
        else:
            proto, _ = prototype_loss_v2(dec, gt, p["prototype_loss.embedding.weight"], th, margin, scale)
    out = F.interpolate(logits, scale_factor=2, mode="bilinear", align_corners=True)   # UpsamplingBilinear2d
    return out, loss, torch.stack(usage) if usage else torch.zeros(0), proto, {"indices": indices, "decoder_out": dec}


def unet_forward(p: Dict[str, Tensor], x: Tensor, training: bool, eps: float = 1e-5, momentum: float = 0.1) -> Tensor:
    """Unet.forward, unet/net.py:833-838: zero-padded encoder (make_encoder default, :820),
    SegmentationHead = 3x3 conv with bias + x2 bilinear (align_corners=True) (segmentation_head.py:78-83)."""
    feats = resnet_encoder(p, x, training, reflect=False, eps=eps, momentum=momentum)[1:]
    dec = unet_decoder(p, feats, training, eps=eps, momentum=momentum)
    y = F.conv2d(dec, p["segmentation_head.0.weight"], p["segmentation_head.0.bias"], padding=1)
    return F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
