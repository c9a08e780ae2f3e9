"""`BaseDataset` with the reference's folder layout, constructor and sample dictionaries (data/dataset.py:15-62).

    <data_dir>/input/<name>.png     RGB images
    <data_dir>/target/<name>.png    grey masks (pixel values 0 / 128 / 255 -> classes through utils.seg_tools.img_to_label)
    split 'labelled'   = the files that have a target;  'unlabelled' = input files without one (:29-35)
    batch_size          pads the file list with its first entries to a multiple of the batch (:38-39)
    resize              int or (w, h): images bilinear, masks nearest (only when target_resize) (:52-55)
    sample              {'filename', 'img' float32 (3, H, W) in [0, 1], ['target' uint8 (H, W)]} -- no mean/std normalisation (q14)

Differences, on purpose: the file order is SORTED (the reference takes os.listdir / a set difference, whose order is not
reproducible: q18), and `TF.to_tensor` of torchvision (absent here) is restated as uint8 HWC -> float32 CHW / 255.
`write_synthetic_dataset` writes a CWFID-shaped folder of synthetic crop / weed blobs (no dataset can be fetched).
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


class BaseDataset(Dataset):
    def __init__(self, data_dir: str, split: str, batch_size: int = None, resize=None, target_resize: bool = True):
        super().__init__()
        if type(resize) == int:
            self.resize = (resize, resize)
        elif type(resize) in (tuple, list):
            self.resize = tuple(resize)
        elif resize is None:
            self.resize = None
        else:
            raise ValueError(f"It's invalid type of resize {type(resize)}")
        self.img_dir = os.path.join(data_dir, "input")
        self.target_resize = target_resize
        targets = sorted(os.listdir(os.path.join(data_dir, "target")))
        if split == "labelled":
            self.filenames, self.target_dir = targets, os.path.join(data_dir, "target")
        elif split == "unlabelled":
            self.filenames, self.target_dir = sorted(set(os.listdir(self.img_dir)) - set(targets)), None
        else:
            raise ValueError("split has to be labelled or unlabelled")
        if batch_size is not None and len(self.filenames) % batch_size != 0:
            self.filenames = self.filenames + self.filenames[0:batch_size - len(self.filenames) % batch_size]

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, index):
        filename = self.filenames[index]
        img = Image.open(os.path.join(self.img_dir, filename)).convert("RGB")
        target = Image.open(os.path.join(self.target_dir, filename)).convert("L") if self.target_dir is not None else None
        if self.resize is not None:
            img = img.resize(self.resize, resample=Image.BILINEAR)
            if self.target_resize and target is not None:
                target = target.resize(self.resize, resample=Image.NEAREST)
        img = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)   # TF.to_tensor
        if target is None:
            return {"filename": filename, "img": img}
        return {"filename": filename, "img": img, "target": torch.from_numpy(np.array(target))}


def write_synthetic_dataset(data_dir: str, n_labelled: int, n_unlabelled: int, size: int = 64, seed: int = 0, cell: int = 8,
                            pixel_values=(0, 128, 255)):
    """A CWFID-shaped folder of synthetic crop / weed blobs: class k's mask pixels carry pixel_values[k] (the reference configs'
    `pixel_to_label` {"0": 0, "128": 1, "255": 2}), images are the class colours + noise (same recipe as trainer.SyntheticCropWeed)."""
    rng = np.random.default_rng(seed)
    palette = np.array([[0.25, 0.20, 0.15], [0.20, 0.55, 0.25], [0.55, 0.60, 0.20]])
    os.makedirs(os.path.join(data_dir, "input"), exist_ok=True)
    os.makedirs(os.path.join(data_dir, "target"), exist_ok=True)
    for i in range(n_labelled + n_unlabelled):
        low = rng.integers(0, len(pixel_values), (max(size // cell, 1),) * 2)
        lab = np.kron(low, np.ones((cell, cell), dtype=np.int64))[:size, :size]
        img = np.clip(palette[lab] + 0.15 * rng.random((size, size, 3)), 0, 1)
        name = f"img_{i:04d}.png"
        Image.fromarray((img * 255).astype(np.uint8)).save(os.path.join(data_dir, "input", name))
        if i < n_labelled:
            Image.fromarray(np.array(pixel_values, dtype=np.uint8)[lab]).save(os.path.join(data_dir, "target", name))
