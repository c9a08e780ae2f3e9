"""UNet decoder (reference: models/networks/unet/decoder.py:7-39).

Five blocks of two fused Conv3x3(pad 1, zeros, no bias)-BatchNorm-ReLU units; between blocks the
output is resized bilinearly (align_corners=False) to the next skip's size and concatenated
(upsampled first, skip second).  Parameter holders are real nn.Conv2d / nn.BatchNorm2d so that
`models.init_weight` and the reference's checkpoint keys (`decoder.blocks.i.j.k.*`) keep working.
"""
from torch import nn

from .... import nnf as _nnf


class ConvBNReLU(nn.Sequential):
    """Sequential(Conv2d, BatchNorm2d, ReLU) parameter layout, fused execution."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, padding=int((kernel_size - 1) / 2), bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU())

    def forward(self, x, x2=None):
        return _nnf.conv_bn_act(x, self[0], self[1], relu=True, x2=x2)


def conv_bn_relu(in_channels: int, out_channels: int, kernel_size: int = 3):
    return ConvBNReLU(in_channels, out_channels, kernel_size)


def double_conv_block(in_channels: int, out_channels: int, kernel_size: int = 3):
    return nn.Sequential(conv_bn_relu(in_channels, out_channels, kernel_size),
                         conv_bn_relu(out_channels, out_channels, kernel_size))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]                     # deep -> shallow, e.g. (2048, 1024, 512, 256, 64)
        blocks, prev = [], 0
        for i, out_ch in enumerate(decoder_channels):
            blocks.append(double_conv_block(enc[i] + prev, out_ch))
            prev = out_ch
        self.blocks = nn.ModuleList(blocks)

    def forward(self, *features):
        dt = features[0].dtype                                      # activation dtype of the un-quantised skips
        feats = [_nnf.cast_act(f, dt) for f in features[::-1]]      # (the VQ layer always returns fp32, vq_img.py:229)
        if any(isinstance(f, _nnf.S3) for f in feats):              # split-3 eval forward: the quantised levels join it
            feats = [_nnf.to_s3(f) for f in feats]
        out = self.blocks[0][1](self.blocks[0][0](feats[0]))
        for i in range(1, len(self.blocks)):
            up = _nnf.upsample_bilinear(out, size=feats[i].shape[-2:], align_corners=False)
            # torch.cat((up, skip), 1) of the reference (decoder.py:35-37) is fused into the conv loader
            out = self.blocks[i][1](self.blocks[i][0](up, feats[i]))
        return out
