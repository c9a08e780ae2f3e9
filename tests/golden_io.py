"""Load golden fixtures (tests/golden/*.npz) captured from the reference by oracle/make_golden.py."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture(dict):
    meta: dict


def load(name: str) -> Fixture:
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    fx = Fixture()
    for k in z.files:
        if k == "meta":
            continue
        a = z[k]
        fx[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a
    fx.meta = json.loads(str(z["meta"]))
    return fx


def layout(name: str):
    with open(os.path.join(GOLDEN, f"state_dict_layout_{name}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def probe(t: torch.Tensor, limit: int = 16384, take: int = 4096) -> torch.Tensor:
    """Same sampling rule as oracle/make_golden.py::probe."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= limit:
        return t.detach()
    stride = flat.numel() // take
    return flat[::stride][:take].clone()
