"""Functional building blocks of the encoder/decoder (the calls the nn.Modules make).

Each function is the single implementation of its op for the accelerated path; tensors are
channels_last (NHWC) on the device.  STATUS (round 1): the VQ layer runs on hand-written HIP
kernels; the convolution / batch-norm / resampling ops below still enqueue PyTorch-ROCm
(MIOpen / ATen) device kernels and are the next ops to move behind include/vqseg.h
(DESIGN.md, section "Kernel inventory and status").  There is no CPU path here either way:
the model refuses CPU tensors at its entry.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def conv2d(x, weight, bias=None, stride=1, padding=0, reflect=False):
    """Conv2d with zero or reflect padding (nn.Conv2d(padding_mode=...) semantics)."""
    if reflect and padding > 0:
        x = F.pad(x, (padding, padding, padding, padding), mode="reflect")
        padding = 0
    return F.conv2d(x, weight, bias, stride=stride, padding=padding)


def batch_norm(x, bn, training):
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)


def conv_bn_act(x, conv, bn, training, relu=True, residual=None):
    """Conv (no bias) -> BatchNorm2d (batch statistics in training mode, running-stat update as
    nn.BatchNorm2d) -> [+ residual] -> [ReLU]."""
    y = conv2d(x, conv.weight, None, conv.stride[0], conv.padding[0], conv.padding_mode == "reflect")
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    y = batch_norm(y, bn, training)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def max_pool_3x3_s2(x):
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def upsample_bilinear(x, size=None, scale_factor=None, align_corners=False):
    return F.interpolate(x, size=size, scale_factor=scale_factor, mode="bilinear", align_corners=align_corners)


def concat_channels(a, b):
    return torch.cat((a, b), dim=1)
