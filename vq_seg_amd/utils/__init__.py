from .load_config import AttrDict, get_config_from_json, get_config_from_yaml  # noqa: F401
from .lr_schedulers import CosineAnnealingLR, WarmUpPolyLR  # noqa: F401
from .seg_tools import img_to_label, onehot_1d, label_to_onehot  # noqa: F401
from .seed import seed_everything  # noqa: F401
from .device import device_setting  # noqa: F401
from .ckpoints import save_ckpoints, load_ckpoints  # noqa: F401
