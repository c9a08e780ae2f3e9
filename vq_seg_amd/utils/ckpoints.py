"""Checkpoint I/O with the reference's dictionary layout (utils/ckpoints.py:7-21), plus the VQ /
prototype `initted` flags so a resumed training run does not re-run k-means."""
import torch


def _initted_flags(model):
    return {name: bool(m.initted) for name, m in model.named_modules() if hasattr(m, "initted")}


def save_ckpoints(model_1, model_2, epoch, batch_idx, optimizer_1, optimizer_2, filepath, models=None):
    blob = {"model_1": model_1, "model_2": model_2, "epoch": epoch, "batch_idx": batch_idx,
            "optimizer_1": optimizer_1, "optimizer_2": optimizer_2}
    if models is not None:
        blob["initted"] = [_initted_flags(m) for m in models]
    torch.save(blob, filepath)


def load_ckpoints(weights_path, istrain: bool, map_location=None):
    ck = torch.load(weights_path, map_location=map_location, weights_only=True)
    if istrain:
        return ck["model_1"], ck["model_2"], ck["epoch"], ck["batch_idx"], ck["optimizer_1"], ck["optimizer_2"], \
            ck.get("initted")
    return ck.get("model_1", ck)


def restore_initted(model, flags):
    if not flags:
        return
    mods = dict(model.named_modules())
    for name, v in flags.items():
        if name in mods:
            mods[name].initted = bool(v)
