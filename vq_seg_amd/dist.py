"""Thin torch.distributed helpers (RCCL on the GPU, gloo in the CPU tests).

One process per GPU.  The hot path shards the minibatch: the only collectives
are (1) the gradient all-reduce, (2) the k-means-init sums/counts all-reduce plus the
broadcast of rank 0's initial draw.  No collective sits inside the VQ kernels.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def collectives_on() -> bool:
    """True when the data-parallel collectives run: more than one rank, or a single-rank process group with
    VQSEG_DIST_SINGLE=1 -- the way to drive the whole N > 1 code path through RCCL on a one-GPU box (a one-rank communicator:
    every collective is the identity, but initialisation, the averaging reduction, async work handles and stream ordering are
    the real thing)."""
    return world_size() > 1 or (is_dist() and os.environ.get("VQSEG_DIST_SINGLE") == "1")


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def broadcast0(t: torch.Tensor) -> torch.Tensor:
    if collectives_on():
        dist.broadcast(t, src=0)
    return t


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    if collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
