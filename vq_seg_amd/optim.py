"""The optimiser step of the training path as one HIP launch per network (csrc/optim_kernels.hip).

Reference: `torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.999))` (train_vqreptunet1x1v2.py:106-107), stepped at :200-201.
`HipAdam` IS a torch.optim.Adam -- same constructor defaults, same `state` / `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq`
per parameter), so checkpoints written by the reference's optimiser load into it and vice versa -- whose `step()` runs
`vqseg_adam_step_f32`: the non-fused single-tensor arithmetic of torch/optim/adam.py in fp32 over ALL parameters in one launch,
and in the same pass the kernel-side bf16 images of the k x k convolution weights (nnf._pack_all's forward / data-gradient /
split-3 images) are rewritten from the updated values.  Round 3 re-packed every weight lazily after each step: one launch per
layer, 4.9 ms per training step.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import _hip
from ._hip import _check, lib

_REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("numel", "<i8"), ("k", "<i4"), ("cout", "<i4"),
                 ("cin", "<i4"), ("c1", "<i4"), ("fwd", "<u8"), ("tr", "<u8"), ("s3", "<u8")])      # == struct VqsegAdamParam
assert _REC.itemsize == 80


def _image_plan(p: torch.Tensor):
    """(k, {kind: n_elems}, c1) for a convolution weight whose images this step should rewrite, else None (plain parameter)."""
    kinds = getattr(p, "_vq_kinds", None)
    if not kinds or p.dim() != 4 or p.shape[2] != p.shape[3] or p.shape[2] not in (1, 3) or not p.is_contiguous():
        return None
    cout, cin, k, _ = p.shape
    s3 = [kk for kk in kinds if isinstance(kk, tuple)]
    if len(s3) > 1 or (s3 and (cin % 32 or s3[0][1] % 32)):
        return None                                          # nnf._pack_all does not serve these weights either
    cin_p, cout_p = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
    sizes = {}
    if "fwd" in kinds:
        sizes["fwd"] = cout * k * k * cin_p
    if "tr" in kinds:
        sizes["tr"] = cin * k * k * cout_p
    if s3:
        sizes[s3[0]] = cout * k * k * 3 * cin
    return k, sizes, (s3[0][1] if s3 else cin)


class HipAdam(torch.optim.Adam):
    """torch.optim.Adam (defaults of the reference's call: eps 1e-8, weight_decay 0, amsgrad False) on the HIP kernel."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)
        self._tables: Dict[int, dict] = {}

    # -- the launch table of one parameter group at one step count: rebuilt only when a pointer or an image set changes
    def _table(self, gi: int, plist: List[torch.Tensor], exp_avgs, exp_avg_sqs):
        sig = []
        plans = []
        for p, m, v in zip(plist, exp_avgs, exp_avg_sqs):
            plan = _image_plan(p)
            plans.append(plan)
            sig.append((p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), None if plan is None else tuple(sorted(map(str, plan[1])))))
        sig = tuple(sig)
        tab = self._tables.get(gi)
        if tab is not None and tab["sig"] == sig:
            return tab
        L = lib()
        dev = plist[0].device
        rec = np.zeros(len(plist), dtype=_REC)
        items = []
        images = []
        for i, (p, m, v, plan) in enumerate(zip(plist, exp_avgs, exp_avg_sqs, plans)):
            for t, name in ((p, "parameter"), (p.grad, "gradient"), (m, "exp_avg"), (v, "exp_avg_sq")):
                _hip.tptr(t, name, dtype=torch.float32, numel=p.numel())
                if not t.is_contiguous():
                    raise _hip.HipLibraryError(f"HipAdam: {name} of a {tuple(p.shape)} parameter is not contiguous")
            _hip._raise_if_not_on_gpu()
            r = rec[i]
            r["p"], r["g"], r["m"], r["v"], r["numel"] = p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
            imgs = None
            if plan is not None:
                k, sizes, c1 = plan
                bufs = getattr(p, "_vq_img_bufs", None)
                if bufs is None:
                    bufs = p._vq_img_bufs = {}
                imgs = {}
                for kind, n in sizes.items():
                    b = bufs.get(kind)
                    if b is None or b.numel() != n or b.device != dev:
                        b = bufs[kind] = torch.empty(n, dtype=torch.int16, device=dev)
                    imgs[kind] = b
                    r["fwd" if kind == "fwd" else "tr" if kind == "tr" else "s3"] = b.data_ptr()
                r["k"], r["cout"], r["cin"], r["c1"] = k, p.shape[0], p.shape[1], c1
            images.append(imgs)
            n_items = L.vqseg_adam_work_items(p.numel(), int(r["k"]), int(r["cout"]), int(r["cin"]))
            items.append(np.stack([np.full(n_items, i, dtype=np.int32), np.arange(n_items, dtype=np.int32)], axis=1))
        items = np.concatenate(items, axis=0)
        # longest work items first: the 3x3 tiles (9216 elements) before the flat chunks (4096) and the 1x1 tiles
        order = np.argsort(-np.where(rec["k"][items[:, 0]] == 3, 3, np.where(rec["k"][items[:, 0]] == 0, 2, 1)), kind="stable")
        items = np.ascontiguousarray(items[order])
        tab = {"sig": sig, "n_items": int(items.shape[0]), "images": images,
               "rec": torch.from_numpy(rec.view(np.uint8).copy()).to(dev), "items": torch.from_numpy(items).to(dev)}
        self._tables[gi] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        for gi, group in enumerate(self.param_groups):
            if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"] or group.get("capturable") or group.get("differentiable"):
                raise NotImplementedError("HipAdam implements the reference's call: Adam(params, lr, betas) with the defaults")
            plist: List[torch.Tensor] = []
            grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], []
            self._init_group(group, plist, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not plist:
                continue
            for i, p in enumerate(plist):                    # a checkpoint of a fused / capturable optimiser holds device-side counters
                if steps[i].device.type != "cpu":
                    steps[i] = self.state[p]["step"] = steps[i].detach().cpu()
            for t in steps:
                t += 1                                       # CPU scalars (torch's own layout of a non-capturable step)
            by_step: Dict[int, List[int]] = {}
            for i, t in enumerate(steps):
                by_step.setdefault(int(t.item()), []).append(i)
            beta1, beta2 = group["betas"]
            lr = group["lr"]
            lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
            for step, idx in by_step.items():
                sub = (gi, step) if len(by_step) > 1 else gi
                ps = plist if len(by_step) == 1 else [plist[i] for i in idx]
                ms = exp_avgs if len(by_step) == 1 else [exp_avgs[i] for i in idx]
                vs = exp_avg_sqs if len(by_step) == 1 else [exp_avg_sqs[i] for i in idx]
                tab = self._table(sub, ps, ms, vs)
                with _hip.on_device(ps[0].device):
                    _check(L.vqseg_adam_step_f32(tab["rec"].data_ptr(), tab["items"].data_ptr(), tab["n_items"], lr, float(beta1), float(beta2),
                                                 float(group["eps"]), step, _hip._stream()), "vqseg_adam_step_f32")
                for p, imgs in zip(ps, tab["images"]):
                    if imgs is not None:
                        # the images ARE those of the new values: install them (the post-step hook of _wcache keeps a "fresh" cache)
                        p._vq_pack = {"key": (p._version, p.data_ptr(), str(p.device)), "all": imgs, "fresh": True}
                    elif getattr(p, "_vq_pack", None) is not None:
                        p._vq_pack = None
        return loss
