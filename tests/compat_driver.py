"""Child process of tests/test_compat_gpu.py: the reference trainer's situation on the GPU box.

    python tests/compat_driver.py <repo>/compat <out.npz> iter_v1|iter_v2|iter_v1_amp|iter_v2_amp|curve|curve_bf16|curve_amp|curve128|curve128_bf16|curve128_amp|train_main

sys.path gets `<repo>/compat` (the maintainer's one line) and -- for the restated loop body only, which lives under tests/ --
the repository root.  Models, losses, metrics and the schedule are reached through the reference's TOP-LEVEL names
(`import models`, `from loss import make_loss`, ...), never through `vq_seg_amd.` paths.
"""
import os
import sys
import types

compat_dir, out_path, what = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, compat_dir)
sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                   # noqa: E402
import torch                                         # noqa: E402

import models                                        # noqa: E402  (train_vqreptunet1x1v2.py:13)
from utils.load_config import EasyDict               # noqa: E402  (the reference wraps its JSON config in easydict.EasyDict)
from utils.lr_schedulers import CosineAnnealingLR    # noqa: E402
from loss import make_loss                           # noqa: E402
from measurement import Measurement                  # noqa: E402

from tests import cases, cps_loop                    # noqa: E402

assert torch.cuda.is_available(), "needs the MI355X"
ns = types.SimpleNamespace(models=models, make_loss=make_loss, Measurement=Measurement, CosineAnnealingLR=CosineAnnealingLR)
dev = torch.device("cuda:0")


def prepare(model, x, gt, version):
    cases.prepare_module_model(model, x, gt, version, to_input=lambda t: t.contiguous(memory_format=torch.channels_last))


arrays = {}
def _flatten(outs, prefix=""):
    for i, o in enumerate(outs):
        for k, v in o.items():
            if k.startswith("grad_none"):
                arrays[f"{prefix}it{i}/{k}"] = np.array(v)
            elif k.startswith("param/"):
                arrays[prefix + k] = v.numpy()
            elif isinstance(v, float):
                arrays[f"{prefix}it{i}/{k}"] = np.array(v, dtype=np.float64)
            else:
                arrays[f"{prefix}it{i}/{k}"] = v.numpy()


if what in ("iter_v1", "iter_v2"):
    _flatten(cps_loop.run_iterations(ns, int(what[6]), dev, n_iters=2, backward=True, to_cfg=EasyDict, prepare=prepare))
elif what in ("iter_v1_amp", "iter_v2_amp"):
    # the trainer's LITERAL mixed-precision region: torch.cuda.amp.autocast(enabled=True) (no dtype: float16) + GradScaler -- and,
    # beside it, the same iterations under an explicit bfloat16 autocast without a scaler (what this repository's modules compute in
    # ANY enabled autocast region): the two must agree
    _flatten(cps_loop.run_iterations(ns, int(what[6]), dev, n_iters=2, backward=True, to_cfg=EasyDict, prepare=prepare, half=True, scaler=True), "amp/")
    _flatten(cps_loop.run_iterations(ns, int(what[6]), dev, n_iters=2, backward=True, to_cfg=EasyDict, prepare=prepare, half=True,
                                     amp_dtype=torch.bfloat16), "bf16/")
elif what == "curve":
    arrays = cps_loop.run_curve(ns, dev, to_cfg=EasyDict, prepare=prepare)
elif what == "curve_bf16":                            # the benchmarked precision: training forwards / backwards under bf16 autocast
    arrays = cps_loop.run_curve(ns, dev, to_cfg=EasyDict, prepare=prepare, half=True, amp_dtype=torch.bfloat16)
elif what == "curve_amp":                             # `half: true` as shipped: fp16-default autocast + GradScaler, literally
    arrays = cps_loop.run_curve(ns, dev, to_cfg=EasyDict, prepare=prepare, half=True, scaler=True)
elif what in ("curve128", "curve128_bf16", "curve128_amp"):
    # r4: the thicker mIoU-parity run -- 128x128, K = 512 at the three levels, 200 v1 iterations (fixture cps_curve_v1_128.npz)
    kw = dict(curve128={}, curve128_bf16=dict(half=True, amp_dtype=torch.bfloat16), curve128_amp=dict(half=True, scaler=True))[what]
    arrays = cps_loop.run_curve(ns, dev, to_cfg=EasyDict, prepare=prepare, k=cps_loop.K128, spec=cps_loop.CURVE128, **kw)
elif what == "train_main":
    # train() of train_vqreptunet1x1v2.py:48-218 as far as the hot path goes: datasets + loaders (:86-93), models + init_weight
    # (:70-80, random init, k-means codebook / prototype init in the first training forward), optimisers + schedule + AMP region
    # (:102-114; fp16 -> bf16), the iteration (:129-211) -- all through the flat names, on a synthetic CWFID-shaped folder
    import itertools, tempfile
    import torch.nn as nn
    from torch.utils.data import DataLoader
    from data.dataset import BaseDataset, write_synthetic_dataset
    from utils.seg_tools import img_to_label
    from utils.seed import seed_everything
    seed_everything()
    root = tempfile.mkdtemp()
    write_synthetic_dataset(os.path.join(root, "train"), n_labelled=6, n_unlabelled=10, size=64, seed=1)
    sup_loader = DataLoader(BaseDataset(os.path.join(root, "train"), split="labelled", batch_size=4, resize=64), batch_size=4, shuffle=True)
    unsup_loader = DataLoader(BaseDataset(os.path.join(root, "train"), split="unlabelled", batch_size=4, resize=64), batch_size=4, shuffle=True)
    cfg = EasyDict(cps_loop.model_cfg(2, (0, 0, 32, 32, 32)))
    model_1, model_2 = models.networks.make_model(cfg).to(dev), models.networks.make_model(cfg).to(dev)
    for m in (model_1, model_2):
        models.init_weight([m.decoder, m.segmentation_head], nn.init.kaiming_normal_, nn.BatchNorm2d, 1e-5, 0.1, mode="fan_in", nonlinearity="relu")
    loop = cps_loop.Loop(ns, 2, model_1, model_2, total_iters=len(unsup_loader) * 2, half=True, amp_dtype=torch.bfloat16)
    losses, mious = [], []
    for epoch in range(2):
        for sup_dict, unsup_dict in zip(itertools.cycle(sup_loader), unsup_loader):
            l_target = img_to_label(sup_dict["target"], {"0": 0, "128": 1, "255": 2})
            out = loop.iteration(sup_dict["img"].to(dev), l_target.to(dev), unsup_dict["img"].to(dev))
            losses.append(out["loss"]), mious.append(out["step_miou"])
    arrays = dict(losses=np.array(losses), mious=np.array(mious), initted=np.array([bool(model_1.codebook[i].codebook.initted) for i in (2, 3, 4)]),
                  iters=np.array(len(losses)))
else:
    raise SystemExit(f"unknown job {what}")
arrays["module_of_model"] = np.array(type(models.networks.make_model(EasyDict(cps_loop.model_cfg(1, (0, 0, 8, 8, 8))))).__module__)
np.savez(out_path, **arrays)
print("compat_driver ok", what)
