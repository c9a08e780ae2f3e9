from .prototype import ReliablePrototypeLoss, ReliablePrototypeLossv2  # noqa: F401
from .segmentation_head import SegmentationHead  # noqa: F401
