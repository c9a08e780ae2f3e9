"""Thin torch.distributed helpers (RCCL on the GPU, gloo in the CPU tests).

One process per GPU.  The hot path shards the minibatch: the only collectives
are (1) the gradient all-reduce, (2) the k-means-init sums/counts all-reduce plus the
broadcast of rank 0's initial draw.  No collective sits inside the VQ kernels.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def broadcast0(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        dist.broadcast(t, src=0)
    return t


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
