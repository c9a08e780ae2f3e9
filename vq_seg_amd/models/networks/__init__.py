"""Model factory (reference: models/networks/__init__.py:9-51).  Only the networks on the
north-star path are registered; the reference's 33 other research variants are out of scope
 and raise a KeyError naming what is available."""
from .modified_vqunet import VQRePTUnet1x1, VQRePTUnet1x1v2
from .unet import Unet

network_dict = {
    "unet": Unet,
    "vqreptunet1x1": VQRePTUnet1x1,
    "vqreptunet1x1v2": VQRePTUnet1x1v2,
}


def make_model(model_cfg):
    name = model_cfg["name"] if isinstance(model_cfg, dict) else model_cfg.name
    params = model_cfg["params"] if isinstance(model_cfg, dict) else model_cfg.params
    if name not in network_dict:
        raise KeyError(f"model {name!r} is not on the accelerated path; available: {sorted(network_dict)}")
    return network_dict[name](**params)
