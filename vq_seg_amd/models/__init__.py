"""models package -- mirrors the reference's `models/__init__.py` (init_weight, :7-26)."""
from torch import nn

from . import encoders  # noqa: F401
from . import modules  # noqa: F401
from . import networks  # noqa: F401


def _init_one(feature, init_func, norm_layer, bn_eps, bn_momentum, **kwargs):
    for _, m in feature.named_modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
            init_func(m.weight, **kwargs)
        elif isinstance(m, norm_layer):
            m.eps = bn_eps
            m.momentum = bn_momentum
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def init_weight(module_list, init_func, norm_layer, bn_eps, bn_momentum, **kwargs):
    """Apply `init_func` to every conv weight and reset every `norm_layer` (eps, momentum, w=1, b=0)."""
    for feature in (module_list if isinstance(module_list, list) else [module_list]):
        _init_one(feature, init_func, norm_layer, bn_eps, bn_momentum, **kwargs)
