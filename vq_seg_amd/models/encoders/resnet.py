"""ResNet encoders for the UNet family.

Own implementation of the torchvision-0.14 ResNet body (un-vendored third-party code in the
reference: models/encoders/resnet.py:7,117-120) with the reference's encoder behaviour on top
(:117-190): no avgpool/fc, `conv1` re-created (PyTorch default init), optional reflect padding on
the stem and on every Bottleneck conv (BasicBlock encoders and the projection shortcuts stay
zero-padded, q13), six feature maps returned.  state_dict keys equal torchvision's
(`conv1.weight`, `bn1.*`, `layerL.B.convK.weight`, `layerL.B.bnK.*`, `layerL.B.downsample.{0,1}.*`).
"""
from __future__ import annotations

import torch
from torch import nn

from ... import nnf


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x
        if self.downsample is not None:
            idt = nnf.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        y = nnf.conv_bn_act(x, self.conv1, self.bn1)
        return nnf.conv_bn_act(y, self.conv2, self.bn2, relu=True, residual=idt)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (carries the stride: torchvision v1.5) -> 1x1 (x4), projection shortcut when needed."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # conv1's data gradient absorbs the shortcut branch's gradient of x in its epilogue (nnf.GradLink)
        link = nnf.GradLink() if (torch.is_grad_enabled() and x.requires_grad) else None
        y = nnf.conv_bn_act(x, self.conv1, self.bn1, link_in=link)
        y = nnf.conv_bn_act(y, self.conv2, self.bn2)
        if self.downsample is None:
            return nnf.conv_bn_act(y, self.conv3, self.bn3, relu=True, residual=x, link_out=link)
        # the projection runs AFTER conv2 so that its backward comes before conv1's (autograd runs later nodes first)
        # (a tagged block input -- an encoder feature that also feeds the decoder / a VQ layer -- has the other consumer's gradient
        # absorbed by this projection's data gradient: nnf fan-in fusion)
        idt = nnf.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False,
                              link_x=link if nnf.py_opt("py_link_projection", 1) else None, absorb_fanin=True)
        return nnf.conv_bn_act(y, self.conv3, self.bn3, relu=True, residual=idt)


resnet_encoders = {
    "resnet18": {"params": {"out_channels": (3, 64, 64, 128, 256, 512), "block": BasicBlock, "layers": [2, 2, 2, 2]}},
    "resnet34": {"params": {"out_channels": (3, 64, 64, 128, 256, 512), "block": BasicBlock, "layers": [3, 4, 6, 3]}},
    "resnet50": {"params": {"out_channels": (3, 64, 256, 512, 1024, 2048), "block": Bottleneck, "layers": [3, 4, 6, 3]}},
    "resnet101": {"params": {"out_channels": (3, 64, 256, 512, 1024, 2048), "block": Bottleneck, "layers": [3, 4, 23, 3]}},
    "resnet152": {"params": {"out_channels": (3, 64, 256, 512, 1024, 2048), "block": Bottleneck, "layers": [3, 8, 36, 3]}},
}


class ResNetEncoder(nn.Module):
    def __init__(self, out_channels, block, layers, depth=5, in_channels=3, padding_mode="zeros"):
        super().__init__()
        self._depth, self._out_channels, self._in_channels = depth, out_channels, in_channels
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        for m in self.modules():                                  # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        # resnet.py:122-125: the stem is re-created, so it carries PyTorch's default conv init
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False, padding_mode=padding_mode)
        if padding_mode != "zeros":
            assert padding_mode in ("reflect",), f"padding_mode {padding_mode} is not available on the accelerated path"
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in layer:
                    if isinstance(blk, Bottleneck):               # resnet.py:143-148: Bottleneck convs only
                        for conv in (blk.conv1, blk.conv2, blk.conv3):
                            conv.padding_mode = padding_mode

    def _make_layer(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, 0, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        """-> [x, stem (C64, /2), layer1 (/4), layer2 (/8), layer3 (/16), layer4 (/32)][: depth + 1]"""
        feats = [x]
        y = nnf.fanin_tag(nnf.stem_conv_bn_act(x, self.conv1, self.bn1))      # two consumers: the pool and the decoder's last block
        feats.append(y)
        y = nnf.max_pool_3x3_s2(y)
        layers = (self.layer1, self.layer2, self.layer3, self.layer4)
        for li, layer in enumerate(layers):
            for blk in layer:
                y = blk(y)
            nxt = layers[li + 1][0] if li + 1 < len(layers) else None
            if isinstance(nxt, Bottleneck) and nxt.downsample is not None and li + 1 <= self._depth - 1:
                nnf.fanin_tag(y)                                  # the next stage's projection absorbs the decoder-side gradient
            feats.append(y)
        return feats[: self._depth + 1]

    def load_state_dict(self, state_dict, **kwargs):
        state_dict = dict(state_dict)
        state_dict.pop("fc.bias", None)
        state_dict.pop("fc.weight", None)
        return super().load_state_dict(state_dict, **kwargs)

    def out_channels(self):
        return self._out_channels[: self._depth + 1]
