"""Top-level names of the reference trainer -> this repository's package.

The reference (chaeyeongyun/VQ_SEG) is a flat source tree: its trainers do `import models`,
`from vector_quantizer import make_vq_module`, `from loss import make_loss`, `from measurement import Measurement`,
`from utils.ckpoints import ...` (train_vqreptunet1x1v2.py:13-26; models/networks/modified_vqunet/net.py imports
`vector_quantizer` the same way).  This repository's implementation lives in ONE package, `vq_seg_amd`, with
package-relative imports.  Putting THIS directory (`<repo>/compat`) in front of `sys.path` makes those flat names
resolve to `vq_seg_amd.*`: every shim package here registers the real module objects -- the package and all of its
submodules -- under the flat names in `sys.modules`, so `models.networks.make_model is vq_seg_amd.models.networks.make_model`
(one set of classes, one copy of every module; `isinstance` checks and `state_dict` keys are unaffected).
"""
import importlib
import os
import pkgutil
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bind(flat_name: str, real_name: str):
    """Register `real_name` (a module or package of vq_seg_amd) and everything below it as `flat_name[...]`."""
    if importlib.util.find_spec("vq_seg_amd") is None:
        sys.path.insert(1, _REPO)                      # the repository root holds the vq_seg_amd package
    real = importlib.import_module(real_name)
    if hasattr(real, "__path__"):                      # import every submodule so that `from utils.seed import ...` finds it
        for info in pkgutil.walk_packages(real.__path__, real_name + "."):
            importlib.import_module(info.name)
    for name, mod in list(sys.modules.items()):
        if mod is not None and (name == real_name or name.startswith(real_name + ".")):
            sys.modules[flat_name + name[len(real_name):]] = mod
    return real
