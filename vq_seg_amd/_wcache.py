"""Kernel-side images of parameters (packed bf16 convolution weights, the VQ kernels' prepared codebook) and WHEN they die.

An image is cached on the Parameter object and is valid for one (version counter, storage pointer, device).  The version
counter alone is NOT enough: `torch.optim.Adam(fused=True).step()` rewrites the parameters without bumping it (measured on
torch 2.10: `_version` stays put across a fused step), and neither do writes through `.data` (the reference's own idiom
`embedding.weight.data.copy_(...)`, vq_img.py:185) nor `dist.broadcast(p.data)`.  So:
  * every optimiser step drops the images of that optimiser's parameters (a global `register_optimizer_step_post_hook`,
    which also fires under `GradScaler.step(optimizer)` -- the reference trainers' call) -- except the images `optim.HipAdam`
    has just rewritten from the updated values inside its own step (it marks that cache "fresh" for exactly one hook call);
  * code that writes parameter storage behind autograd's back calls `invalidate(...)` (this package does so after its own
    broadcast / k-means / EMA writes); `load_state_dict` and ordinary in-place ops bump the version counter and need nothing.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.optim.optimizer import register_optimizer_step_post_hook


def cache_of(weight: torch.Tensor) -> dict:
    """The dictionary of kernel-side images of `weight` (a Parameter), emptied whenever the parameter may have changed."""
    key = (weight._version, weight.data_ptr(), str(weight.device))
    cache = getattr(weight, "_vq_pack", None)
    if cache is None or cache["key"] != key:
        cache = {"key": key}
        weight._vq_pack = cache
    return cache


def invalidate(*objs) -> None:
    """Drop the images of the given parameters / of every parameter of the given modules (call after `.data` writes)."""
    for o in objs:
        params = o.parameters() if isinstance(o, nn.Module) else (o if isinstance(o, (list, tuple)) else [o])
        for p in params:
            if getattr(p, "_vq_pack", None) is not None:
                p._vq_pack = None


def _after_optimizer_step(optimizer, args, kwargs):
    for group in optimizer.param_groups:
        for p in group["params"]:
            cache = getattr(p, "_vq_pack", None)
            if cache is not None:
                if cache.pop("fresh", False):               # optim.HipAdam rewrote the images from the NEW values inside its step
                    continue
                p._vq_pack = None


register_optimizer_step_post_hook(_after_optimizer_step)
