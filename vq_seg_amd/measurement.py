"""Segmentation metrics (reference: measurement.py:7-91).

`Measurement` keeps the reference's numpy interface (the evaluator calls it with numpy arrays);
`confusion_matrix_device` / `miou_device` compute the same quantities on the device from logits and
labels (one HIP pass, vqseg_confusion_counts_f), so the training loop needs neither a per-step `.cpu().numpy()` round
trip nor the host synchronisation torch.bincount implies.
"""
import numpy as np
import torch


class Measurement:
    def __init__(self, num_classes: int, ignore_idx=None):
        self.num_classes, self.ignore_idx = num_classes, ignore_idx

    def _make_confusion_matrix(self, pred: np.ndarray, target: np.ndarray):
        assert pred.shape[0] == target.shape[0], "pred and target ndarray's batchsize must have same value"
        n = pred.shape[0]
        cats = self.num_classes * target.reshape(n, -1) + pred.argmax(axis=1).reshape(n, -1)
        conf = np.stack([np.bincount(row, minlength=self.num_classes ** 2) for row in cats])
        return conf.reshape(n, self.num_classes, self.num_classes)

    def accuracy(self, pred, target):
        lab = pred.argmax(axis=1).reshape(pred.shape[0], -1)
        tgt = target.reshape(target.shape[0], -1)
        if self.ignore_idx is not None:
            keep = np.where(tgt != self.ignore_idx)
            lab, tgt = lab[keep], tgt[keep]
        return np.mean(np.sum(lab == tgt, axis=-1) / lab.shape[-1])

    def miou(self, conf_mat: np.ndarray):
        col, row = np.sum(conf_mat, -2), np.sum(conf_mat, -1)
        ious = [np.mean(conf_mat[:, i, i] / (col[:, i] + row[:, i] - conf_mat[:, i, i] + 1e-8))
                for i in range(self.num_classes)]
        return np.mean(np.array(ious)), ious

    def precision(self, conf_mat):
        col = np.sum(conf_mat, -2)
        per = np.mean(np.array([conf_mat[:, i, i] / (col[:, i] + 1e-7) for i in range(self.num_classes)]), axis=-1)
        return np.mean(per), per

    def recall(self, conf_mat):
        row = np.sum(conf_mat, -1)
        per = np.mean(np.array([conf_mat[:, i, i] / row[:, i] for i in range(self.num_classes)]), axis=-1)
        return np.mean(per), per

    def f1score(self, recall, precision):
        return 2 * recall * precision / (recall + precision)

    def measure(self, pred, target):
        conf = self._make_confusion_matrix(pred, target)
        acc = self.accuracy(pred, target)
        miou, ious = self.miou(conf)
        precision, _ = self.precision(conf)
        recall, _ = self.recall(conf)
        return acc, miou, ious, precision, recall, self.f1score(recall, precision)

    __call__ = measure


def confusion_matrix_device(logits: torch.Tensor, target: torch.Tensor, num_classes: int) -> torch.Tensor:
    """(N, C, H, W) logits, (N, H, W) int labels -> (N, C, C) int64 counts, rows = ground truth."""
    n = logits.shape[0]
    if logits.is_cuda and logits.dim() == 4 and 2 <= num_classes <= 4 and logits.shape[1] == num_classes:
        from . import _hip
        x = logits.detach()
        x = x if x.dtype == torch.float32 else x.float()
        b, c, h, w = x.shape
        sb, sc, sh, sw = x.stride()
        if sh != w * sw:
            x = x.contiguous()
            sb, sc, sh, sw = x.stride()
        tgt = target.reshape(b, h * w).long().contiguous()
        out = torch.empty((b, c, c), dtype=torch.int64, device=x.device)
        with torch.cuda.device(x.device):
            from .nnf import _strided_f32
            _hip._check(_hip.lib().vqseg_confusion_counts_f(_strided_f32(x, "logits", b * c * h * w), sb, sc, sw,
                                                            _hip.tptr(tgt, "targets", dtype=torch.int64, numel=b * h * w), b, c, h * w,
                                                            _hip.tptr(out, "confusion counts", dtype=torch.int64, numel=b * c * c),
                                                            _hip._stream()), "vqseg_confusion_counts_f")
        return out
    cats = num_classes * target.reshape(n, -1).long() + logits.argmax(dim=1).reshape(n, -1)
    cats = cats + (num_classes ** 2) * torch.arange(n, device=cats.device)[:, None]
    return torch.bincount(cats.reshape(-1), minlength=n * num_classes ** 2).reshape(n, num_classes, num_classes)


def miou_device(conf: torch.Tensor):
    """Measurement.miou on the device: per-class IoU averaged over the batch, then over classes."""
    conf = conf.double()
    diag = torch.diagonal(conf, dim1=-2, dim2=-1)
    ious = (diag / (conf.sum(-2) + conf.sum(-1) - diag + 1e-8)).mean(dim=0)
    return ious.mean(), ious
