"""Checkpoint I/O with the reference's dictionary layout and call signatures (utils/ckpoints.py:7-21).

`save_ckpoints` / `load_ckpoints` keep the reference's arity so its callers keep working -- including the quirk
that `load_ckpoints(path, istrain=True)` returns `model_2`'s weights only (the reference never wires
resume).  What the reference cannot do is in separate functions: `load_training_state` returns the whole dictionary
(both models), and the VQ / prototype `initted` flags -- plain attributes the reference does not persist, so that a
resumed training run would re-run k-means over the loaded codebooks -- travel in an extra `initted` key
(`save_ckpoints(..., models=[m1, m2])`, `restore_initted`).  Files are read with `weights_only=True`.

Data parallel runs (trainer.CPSTrainer): parameters and codebooks are bit-identical on all ranks; BatchNorm running
statistics are per-rank (the reference's per-device BatchNorm semantics) -- `CPSTrainer.save_checkpoint()` writes rank 0's
own state without a collective, so saving never changes any rank (`CPSTrainer.sync_buffers()` is the explicit broadcast).
"""
import os
import shutil
import tarfile

import torch


def _initted_flags(model):
    return {name: bool(m.initted) for name, m in model.named_modules() if hasattr(m, "initted")}


def save_ckpoints(model_1, model_2, epoch, batch_idx, optimizer_1, optimizer_2, filepath, models=None, extra=None):
    """model_k / optimizer_k are state_dicts (train_vqreptunet1x1v2.py:245-259).  `models` (optional, the two nn.Modules)
    adds their `initted` flags; `extra` (optional dict of plain values, e.g. the iteration counter of the LR schedule) is stored
    beside the reference's keys."""
    blob = {"model_1": model_1, "model_2": model_2, "epoch": epoch, "batch_idx": batch_idx,
            "optimizer_1": optimizer_1, "optimizer_2": optimizer_2}
    if models is not None:
        blob["initted"] = [_initted_flags(m) for m in models]
    for k, v in (extra or {}).items():
        if k in blob:
            raise KeyError(f"extra key {k!r} collides with the checkpoint layout")
        blob[k] = v
    torch.save(blob, filepath)


def load_ckpoints(weights_path, istrain: bool, map_location=None):
    """utils/ckpoints.py:15-21: istrain -> (model_2, epoch, batch_idx, optimizer_1, optimizer_2), else model_1's weights
    (`test_detailviz.py:90` also accepts a bare state_dict file: `weights.get('model_1', weights)`)."""
    ck = torch.load(weights_path, map_location=map_location, weights_only=True)
    if istrain:
        return ck["model_2"], ck["epoch"], ck["batch_idx"], ck["optimizer_1"], ck["optimizer_2"]
    return ck.get("model_1", ck)


def load_training_state(weights_path, map_location=None) -> dict:
    """The whole checkpoint dictionary: model_1, model_2, epoch, batch_idx, optimizer_1, optimizer_2[, initted]."""
    return torch.load(weights_path, map_location=map_location, weights_only=True)


def restore_initted(model, flags):
    """Apply the `initted` flags saved by save_ckpoints(..., models=...); a checkpoint without them (reference-written)
    leaves the flags alone."""
    if not flags:
        return
    mods = dict(model.named_modules())
    for name, v in flags.items():
        if name in mods:
            mods[name].initted = bool(v)


def save_tar(target_path):
    """utils/ckpoints.py:28-33: pack a run directory into <dir>.tar.gz next to it and remove the directory."""
    head, name = os.path.split(target_path)
    with tarfile.open(os.path.join(head, name + ".tar.gz"), "w:gz") as t:
        t.add(target_path)
    shutil.rmtree(target_path)
