"""Encoder factory (reference: models/encoders/__init__.py:8-32)."""
import os

import torch

from .resnet import ResNetEncoder, resnet_encoders  # noqa: F401


def make_encoder(name: str, in_channels: int = 3, depth: int = 5, weights=None, padding_mode="zeros",
                 output_stride=32, **kwargs):
    if "resnet" not in name or "cca" in name:
        raise NotImplementedError(f"encoder {name!r}: only plain ResNet encoders are on the accelerated path "
                                  "(SURVEY 2 #3: CCA / VGG variants are out of scope)")
    if name not in resnet_encoders:
        raise KeyError(name)
    encoder = ResNetEncoder(depth=depth, **resnet_encoders[name]["params"], in_channels=in_channels,
                            padding_mode=padding_mode, **kwargs)
    if weights is not None:
        # The reference downloads ImageNet / SWSL weights by URL (models/encoders/__init__.py:24-29).
        # There is no network here: accept a local state_dict file, otherwise fail loudly.
        path = weights if os.path.isfile(str(weights)) else None
        if path is None:
            raise RuntimeError(f"encoder_weights={weights!r}: pretrained weights are fetched by URL in the reference; "
                               "pass a local .pth path or encoder_weights=None (random init)")
        encoder.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    if output_stride != 32:
        raise NotImplementedError("dilated encoders (output_stride != 32) are not used by the target configs")
    return encoder
