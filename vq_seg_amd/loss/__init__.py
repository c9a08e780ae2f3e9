"""Loss factory (reference: loss/__init__.py:9-24); dice + cross-entropy are the ones the path uses."""
from torch import nn

from .dice_loss import DiceLoss, dice_loss  # noqa: F401

loss_dict = {"cross_entropy": nn.CrossEntropyLoss, "dice_loss": DiceLoss, "nll_loss": nn.NLLLoss}


def make_loss(loss_name: str, num_classes: int, ignore_index: int = -100, weight=None):
    if loss_name in ("cross_entropy", "nll_loss"):
        return loss_dict[loss_name](ignore_index=ignore_index, weight=weight)
    if loss_name not in loss_dict:
        raise KeyError(f"loss {loss_name!r} is not on the accelerated path; available: {sorted(loss_dict)}")
    return loss_dict[loss_name](num_classes=num_classes, ignore_index=ignore_index, weight=weight)
