"""Child process of tests/test_compat_gpu.py: the reference trainer's situation on the GPU box.

    python tests/compat_driver.py <repo>/compat <out.npz> iter_v1|iter_v2|curve

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
if what in ("iter_v1", "iter_v2"):
    version = int(what[-1])
    outs = cps_loop.run_iterations(ns, version, dev, n_iters=2, backward=True, to_cfg=EasyDict, prepare=prepare)
    for i, o in enumerate(outs):
        for k, v in o.items():
            if k.startswith("grad_none"):
                arrays[f"it{i}/{k}"] = np.array(v)
            elif k.startswith("param/"):
                arrays[k] = v.numpy()
            elif isinstance(v, float):
                arrays[f"it{i}/{k}"] = np.array(v, dtype=np.float64)
            else:
                arrays[f"it{i}/{k}"] = v.numpy()
elif what == "curve":
    arrays = cps_loop.run_curve(ns, dev, to_cfg=EasyDict, prepare=prepare)
else:
    raise SystemExit(f"unknown job {what}")
arrays["module_of_model"] = np.array(type(models.networks.make_model(EasyDict(cps_loop.model_cfg(1, (0, 0, 8, 8, 8))))).__module__)
np.savez(out_path, **arrays)
print("compat_driver ok", what)
