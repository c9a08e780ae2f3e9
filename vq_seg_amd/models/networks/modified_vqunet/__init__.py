from .net import VQRePTUnet1x1, VQRePTUnet1x1v2  # noqa: F401
