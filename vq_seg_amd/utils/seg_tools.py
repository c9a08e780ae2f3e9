"""Label helpers (reference: utils/seg_tools.py:3-34)."""
import torch


def img_to_label(target_img: torch.Tensor, pixel_to_label_dict: dict) -> torch.Tensor:
    """Map raw mask pixel values (e.g. 0/128/255) to class ids; returns int64."""
    out = target_img
    for pixel, label in pixel_to_label_dict.items():
        out = torch.where(out == int(pixel), label, out)
    return out.long()


def label_to_onehot(target: torch.Tensor, num_classes: int, eps: float = 1e-6) -> torch.Tensor:
    if target.dim() == 3:
        target = target.unsqueeze(1)
    shape = (target.shape[0], num_classes, target.shape[2], target.shape[3])
    return torch.zeros(shape, dtype=torch.float64, device=target.device).scatter_(1, target.long(), 1.0) + eps


def onehot_1d(target: torch.Tensor, num_classes: int, eps: float = 1e-6) -> torch.Tensor:
    """(P,) or (P,1) labels -> (P, num_classes) float64 one-hot + eps (f64 as in the reference, q11)."""
    if target.dim() == 1:
        target = target.unsqueeze(-1)
    out = torch.zeros((target.shape[0], num_classes), dtype=torch.float64, device=target.device)
    return out.scatter_(1, target.long(), 1.0) + eps
