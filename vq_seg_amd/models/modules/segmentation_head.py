"""SegmentationHead used by the plain Unet (reference: models/modules/segmentation_head.py:78-83):
Conv2d(k, padding k//2, with bias) -> UpsamplingBilinear2d(scale) (align_corners=True) -> activation."""
from torch import nn

from ... import nnf


class _Up(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale_factor = scale

    def forward(self, x):
        return nnf.upsample_bilinear(x, scale_factor=self.scale_factor, align_corners=True)


class _Conv(nn.Conv2d):
    """3x3 conv WITH bias, 32 -> num_classes: only the plain `unet` plumbing model (BASELINE config #1, a CPU wiring check in the
    reference) has this layer.  r4: on the precise-mode HIP kernels like everything else (output channels padded to their 4-channel
    granule, the bias in the convolution's epilogue: nnf.conv2d_bias); shapes those kernels do not take (input channels not a
    multiple of 4, stride / rectangular padding) raise there."""

    def forward(self, x):
        if self.stride != (1, 1) or self.padding[0] != self.padding[1] or self.dilation != (1, 1) or self.groups != 1:
            raise NotImplementedError("segmentation head: stride 1, square padding, no dilation / groups on the accelerated path")
        return nnf.conv2d_bias(x, self.weight, self.bias, self.padding[0])


class SegmentationHead(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, upsampling=1, activation=nn.Identity):
        conv = _Conv(in_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2)
        up = _Up(upsampling) if upsampling > 1 else nn.Identity()
        act = activation() if activation in (nn.Softmax2d, nn.Identity) else activation(dim=1)
        super().__init__(conv, up, act)
