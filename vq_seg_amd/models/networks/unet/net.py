"""Plain UNet (reference: models/networks/unet/net.py:806-838) -- BASELINE config #1 plumbing model."""
from torch import nn

from ...encoders import make_encoder
from ...modules.segmentation_head import SegmentationHead
from .decoder import UnetDecoder


class Unet(nn.Module):
    def __init__(self, encoder_name: str, num_classes: int, in_channels: int = 3, decoder_channels=None, depth: int = 5,
                 activation=nn.Identity, upsampling=2, encoder_weights=None):
        super().__init__()
        # unet/net.py:820: the encoder is built WITHOUT weights and with zero padding; `encoder_weights` is ignored
        self.encoder = make_encoder(encoder_name, in_channels, depth)
        enc_ch = self.encoder.out_channels()
        if decoder_channels is None:
            decoder_channels = [c // 2 for c in enc_ch[1:]][::-1]
        self.decoder = UnetDecoder(enc_ch, decoder_channels)
        self.segmentation_head = SegmentationHead(in_channels=decoder_channels[-1], out_channels=num_classes,
                                                  upsampling=upsampling, activation=activation, kernel_size=3)

    def forward(self, x):
        feats = self.encoder(_to_device_layout(x))[1:]
        return self.segmentation_head(self.decoder(*feats))

    def freeze_encoder(self):
        for p in self.encoder.parameters():
            p.requires_grad = False


def _to_device_layout(x):
    """Model entry: refuse CPU tensors (no fallback) and switch to channels_last (NHWC)."""
    import torch
    if not x.is_cuda:
        raise RuntimeError("vq_seg_amd models run on a 'cuda' (ROCm) device only; there is no CPU fallback "
                           f"(got a tensor on {x.device})")
    return x.contiguous(memory_format=torch.channels_last)
