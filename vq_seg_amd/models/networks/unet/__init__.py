from .decoder import UnetDecoder, conv_bn_relu, double_conv_block  # noqa: F401
from .net import Unet  # noqa: F401
