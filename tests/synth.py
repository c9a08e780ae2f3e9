"""Deterministic synthetic tensors shared by the golden generator and the tests.

Everything is derived from numpy's PCG64 `random()` (uniform doubles) followed only by
IEEE-exact operations (subtract, multiply by a constant, max), so the same arrays come
out on any machine -- the golden fixtures store OUTPUTS only (plus float64 checksums of
the regenerated inputs), which keeps tests/golden/ small even for the 69 M-parameter model.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch


def uniform(seed: int, shape: Iterable[int], lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(tuple(shape))) if tuple(shape) else 1
    u = np.random.Generator(np.random.PCG64(seed)).random(n)             # float64 in [0,1)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(tuple(shape))


def relu_features(seed: int, shape: Iterable[int], sparsity: float = 0.4, scale: float = 2.0) -> torch.Tensor:
    """Post-ReLU-looking activations (encoder outputs are non-negative)."""
    return torch.clamp(uniform(seed, shape) - sparsity, min=0.0) * scale


def labels(seed: int, shape: Iterable[int], num_classes: int = 3) -> torch.Tensor:
    return torch.floor(uniform(seed, shape) * num_classes).clamp(max=num_classes - 1).long()


def blob_labels(seed: int, b: int, size: int, num_classes: int = 3, cell: int = 16) -> torch.Tensor:
    """Spatially coherent label maps: low-res random classes, nearest-upsampled (learnable)."""
    low = labels(seed, (b, 1, max(size // cell, 1), max(size // cell, 1)), num_classes).float()
    return torch.nn.functional.interpolate(low, size=(size, size), mode="nearest")[:, 0].long()


def checksum(t: torch.Tensor) -> float:
    return float(t.detach().double().sum().item())


def bits_checksum(t: torch.Tensor) -> int:
    """Order-independent, exact checksum of an fp32 tensor: the int64 sum of its bit patterns (a float64 sum of arbitrary floats
    depends on the summation order, i.e. on torch's thread count)."""
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())


def synth_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Fill a state_dict layout (name -> shape) with deterministic, sanely scaled values.

    conv / linear weights: U(-b, b), b = sqrt(6 / fan_in)  (variance 2/fan_in, He-like)
    BN weight U(0.5, 1.5) (the last BN of a bottleneck, `bn3`: U(0.05, 0.25), so that perturbations are
    not amplified ~4x per encoder stage), bias U(-0.1, 0.1), running_mean U(-0.1, 0.1), running_var U(0.5, 1.5)
    codebooks (`...codebook.embedding.weight`): U(0, 0.5)  (features are non-negative)
    prototypes (`prototype_loss.embedding.weight`): U(-1, 1)
    """
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i, (name, shape) in enumerate(shapes.items()):
        s = seed * 100003 + i
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_var"):
            out[name] = uniform(s, shape, 0.5, 1.5)
        elif name.endswith("running_mean"):
            out[name] = uniform(s, shape, -0.1, 0.1)
        elif "codebook.embedding.weight" in name:
            out[name] = uniform(s, shape, 0.0, 0.5)
        elif name.startswith("prototype_loss."):
            out[name] = uniform(s, shape, -1.0, 1.0)
        elif name.endswith(".bn3.weight"):
            out[name] = uniform(s, shape, 0.05, 0.25)            # damped residual branches (zero_init_residual-like):
        elif len(shape) == 1 and name.endswith(".weight"):       # keeps a random-weight ResNet out of the chaotic regime
            out[name] = uniform(s, shape, 0.5, 1.5)
        elif len(shape) == 1:
            out[name] = uniform(s, shape, -0.1, 0.1)
        else:
            fan_in = int(np.prod(shape[1:]))
            b = math.sqrt(6.0 / fan_in)
            out[name] = uniform(s, shape, -b, b)
    return out


def shapes_of(state_dict: Mapping[str, torch.Tensor]) -> "OrderedDict[str, Tuple[int, ...]]":
    return OrderedDict((k, tuple(v.shape)) for k, v in state_dict.items())


def decoder_shapes(encoder_channels, decoder_channels, prefix: str = "decoder") -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict layout of UnetDecoder (reference keys `decoder.blocks.i.j.k.*`)."""
    enc = list(encoder_channels[1:])[::-1]
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    prev = 0
    for i, co in enumerate(decoder_channels):
        ci = enc[i] + prev
        for j, cin in enumerate((ci, co)):
            base = f"{prefix + '.' if prefix else ''}blocks.{i}.{j}"
            shapes[f"{base}.0.weight"] = (co, cin, 3, 3)
            shapes[f"{base}.1.weight"] = (co,)
            shapes[f"{base}.1.bias"] = (co,)
            shapes[f"{base}.1.running_mean"] = (co,)
            shapes[f"{base}.1.running_var"] = (co,)
            shapes[f"{base}.1.num_batches_tracked"] = ()
        prev = co
    return shapes


PALETTE = ((0.25, 0.20, 0.15), (0.20, 0.55, 0.25), (0.55, 0.60, 0.20))     # soil / crop / weed colours


def blob_images(seed: int, b: int, size: int, cell: int = 16, noise: float = 0.15, num_classes: int = 3):
    """Synthetic crop/weed batch: (images (b, 3, size, size) in [0, 1], labels (b, size, size) int64).  Every class paints its
    blobs in one colour, plus uniform noise -- a learnable stand-in for CWFID (no dataset can travel), deterministic on any
    machine like everything else here (`trainer.SyntheticCropWeed` is the same recipe on torch's generator)."""
    lab = blob_labels(seed, b, size, num_classes, cell)
    pal = torch.tensor(PALETTE[:num_classes], dtype=torch.float32)
    img = pal[lab].permute(0, 3, 1, 2) + noise * uniform(seed + 7919, (b, 3, size, size))
    return img.clamp(0.0, 1.0).contiguous(), lab
