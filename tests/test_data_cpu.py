"""Data path: vq_seg_amd.data.BaseDataset keeps the reference's folder layout, split rule, batch padding, resize
and sample dictionaries (data/dataset.py:15-62), checked on a synthetic CWFID-shaped folder; with utils.seg_tools.img_to_label the
samples are exactly what the trainers' loop bodies consume (train_vqreptunet1x1v2.py:130-139)."""
import itertools
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from vq_seg_amd.data import BaseDataset, write_synthetic_dataset
from vq_seg_amd.utils.seg_tools import img_to_label


def test_base_dataset_contract(tmp_path):
    root = str(tmp_path / "train")
    write_synthetic_dataset(root, n_labelled=5, n_unlabelled=7, size=48, seed=3)
    sup = BaseDataset(root, split="labelled", batch_size=4, resize=32)
    unsup = BaseDataset(root, split="unlabelled", batch_size=4, resize=32)
    assert len(sup) == 8 and len(unsup) == 8                                  # 5 -> 8, 7 -> 8: padded with the first entries (:38-39)
    assert sup.filenames[5:] == sup.filenames[:3] and unsup.filenames[7:] == unsup.filenames[:1]
    assert set(sup.filenames).isdisjoint(unsup.filenames)                       # unlabelled = input files without a target (:33)
    assert len(set(sup.filenames)) == 5 and len(set(unsup.filenames)) == 7
    s = sup[0]
    assert set(s) == {"filename", "img", "target"} and set(unsup[0]) == {"filename", "img"}
    assert s["img"].dtype == torch.float32 and s["img"].shape == (3, 32, 32) and 0.0 <= float(s["img"].min()) and float(s["img"].max()) <= 1.0
    assert s["target"].dtype == torch.uint8 and s["target"].shape == (32, 32) and set(s["target"].unique().tolist()) <= {0, 128, 255}
    full = BaseDataset(root, split="labelled", batch_size=1, resize=32, target_resize=False)       # the evaluator's form (test_detailviz.py:54)
    assert full[0]["target"].shape == (48, 48) and full[0]["img"].shape == (3, 32, 32)
    raw = BaseDataset(root, split="labelled")
    assert raw[0]["img"].shape == (3, 48, 48) and len(raw) == 5
    with pytest.raises(ValueError):
        BaseDataset(root, split="validation")
    with pytest.raises(ValueError):
        BaseDataset(root, split="labelled", resize=3.5)
    # the trainers' iteration (train_vqreptunet1x1v2.py:118,130-132): zip(cycle(sup_loader), unsup_loader) + img_to_label
    sup_loader, unsup_loader = DataLoader(sup, batch_size=4, shuffle=False), DataLoader(unsup, batch_size=4, shuffle=False)
    n = 0
    for sup_dict, unsup_dict in zip(itertools.cycle(sup_loader), unsup_loader):
        l_target = img_to_label(sup_dict["target"], {"0": 0, "128": 1, "255": 2})
        assert sup_dict["img"].shape == (4, 3, 32, 32) and unsup_dict["img"].shape == (4, 3, 32, 32)
        assert l_target.dtype == torch.int64 and set(l_target.unique().tolist()) <= {0, 1, 2}
        n += 1
    assert n == len(unsup_loader) == 2
    # images are the class colours + noise: the label is recoverable from the colour (the task is learnable)
    img, lab = raw[0]["img"], img_to_label(raw[0]["target"], {"0": 0, "128": 1, "255": 2})
    pal = torch.tensor([[0.25, 0.20, 0.15], [0.20, 0.55, 0.25], [0.55, 0.60, 0.20]]) + 0.075
    guess = ((img.permute(1, 2, 0)[:, :, None, :] - pal[None, None]) ** 2).sum(-1).argmin(-1)
    assert (guess == lab).float().mean() > 0.95
