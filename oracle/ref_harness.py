"""Import harness for the READ-ONLY reference at /root/reference -- build-container only.

Used solely by `oracle/make_golden.py` to capture golden vectors from the reference's own
modules.  Nothing under tests/, bench.py or the product imports this file, and it cannot
run on the GPU box (the reference does not travel).

What it registers in sys.modules so that `import models`, `import loss`, ... from the
reference tree succeed:
  * import-only stand-ins with NO arithmetic: `easydict` (attribute dict),
    `pretrainedmodels.models.torchvision_models`, `torchvision.transforms*`,
    `torchvision.models.vgg`;
  * `torchvision.models.resnet`: a torchvision-LIKE ResNet/Bottleneck/BasicBlock written
    here from the published torchvision 0.14 semantics.  torchvision itself is absent
    and cannot be installed, so the ResNet body's numerics stay PARITY UNPINNED; what
    this buys is that everything the reference itself wrote on top of that base
    (reflect-padding rewrite, stage split, VQ-UNet glue) runs for real.
"""
from __future__ import annotations

import sys
import types
from collections import defaultdict

import torch
from torch import nn

REF_ROOT = "/root/reference"


class AttrDict(dict):
    """Minimal easydict.EasyDict: recursive attribute access."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(AttrDict(i) if isinstance(i, dict) and not isinstance(i, AttrDict) else i for i in v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


# ---- torchvision-like ResNet (published v0.14 semantics; harness code, not reference code) ----
def _c3(i, o, s=1):
    return nn.Conv2d(i, o, 3, s, 1, bias=False)


def _c1(i, o, s=1):
    return nn.Conv2d(i, o, 1, s, 0, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, **_):
        super().__init__()
        self.conv1, self.bn1 = _c3(inplanes, planes, stride), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _c3(planes, planes), nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(y)) + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, **_):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1, self.bn1 = _c1(inplanes, width), nn.BatchNorm2d(width)
        self.conv2, self.bn2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False), nn.BatchNorm2d(width)
        self.conv3, self.bn3 = _c1(width, planes * 4), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, groups=1, width_per_group=64, **_):
        super().__init__()
        self.inplanes, self.groups, self.base_width = 64, groups, width_per_group
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(_c1(self.inplanes, planes * block.expansion, stride),
                                 nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, down, groups=self.groups, base_width=self.base_width)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes, groups=self.groups, base_width=self.base_width) for _ in range(1, n)]
        return nn.Sequential(*blocks)


_INSTALLED = False


def install():
    """Register the stand-ins and put the reference root on sys.path (idempotent)."""
    global _INSTALLED
    if _INSTALLED:
        return
    sys.dont_write_bytecode = True          # the reference tree is read-only

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("easydict", EasyDict=AttrDict)
    tv = mod("torchvision")
    tv.models = mod("torchvision.models")
    tv.models.resnet = mod("torchvision.models.resnet", ResNet=ResNet, BasicBlock=BasicBlock, Bottleneck=Bottleneck)

    class _VGG(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tv.models.vgg = mod("torchvision.models.vgg", VGG=_VGG, make_layers=lambda *a, **k: nn.Sequential())
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional")
    pm = mod("pretrainedmodels")
    pm.models = mod("pretrainedmodels.models")
    pm.models.torchvision_models = mod("pretrainedmodels.models.torchvision_models",
                                       pretrained_settings=defaultdict(dict))
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _INSTALLED = True


def ref_modules():
    """Import and return the reference modules used for golden capture."""
    install()
    import models  # noqa: F401  (reference package)
    import vector_quantizer.vq_img as vq_img
    import models.networks as networks
    import models.networks.unet.decoder as decoder
    import models.modules.prototype as prototype
    import loss as loss_pkg
    import measurement
    import utils.lr_schedulers as lr_schedulers
    import utils.seg_tools as seg_tools
    return types.SimpleNamespace(models=models, vq_img=vq_img, networks=networks, decoder=decoder,
                                 prototype=prototype, loss=loss_pkg, measurement=measurement,
                                 lr_schedulers=lr_schedulers, seg_tools=seg_tools, AttrDict=AttrDict)
