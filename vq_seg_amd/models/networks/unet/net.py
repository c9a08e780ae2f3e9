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

    def forward_plumbing(self, x):
        """BASELINE configs[0] ("config/CWFID_Unet.json plain UNet 256x256 bs=2 on CPU: plumbing, no VQ, no GPU"): the same network on
        plain torch operators -- every parameter holder's OWN torch forward (nn.Conv2d, nn.BatchNorm2d, nn.MaxPool2d) plus
        F.interpolate / torch.cat -- on whatever device the tensors live on, CPU included.  An EXPLICIT entry point for the wiring check
        (factory, config, loss, metric, optimiser) the reference runs on the CPU; it is NOT a fallback: `forward` keeps refusing CPU
        tensors, nothing on the HIP path calls this, and it never touches `oracle/`.  Same state_dict, same outputs as the reference's
        `Unet.forward` (models/networks/unet/net.py:833-838; tests/test_unet_plumbing_cpu.py against the reference's golden vectors)."""
        import torch
        import torch.nn.functional as F
        enc = self.encoder

        def block(blk, t):
            idt = t if blk.downsample is None else blk.downsample[1](blk.downsample[0](t))
            if hasattr(blk, "conv3"):                        # Bottleneck
                y = F.relu(blk.bn1(blk.conv1(t)))
                y = F.relu(blk.bn2(blk.conv2(y)))
                return F.relu(blk.bn3(blk.conv3(y)) + idt)
            y = F.relu(blk.bn1(blk.conv1(t)))
            return F.relu(blk.bn2(blk.conv2(y)) + idt)

        y = F.relu(enc.bn1(enc.conv1(x)))
        feats = [y]
        y = enc.maxpool(y)
        for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
            for blk in layer:
                y = block(blk, y)
            feats.append(y)
        feats = feats[: enc._depth][::-1]                    # deep -> shallow

        def cbr(m, t):                                       # ConvBNReLU = Sequential(Conv2d, BatchNorm2d, ReLU)
            return F.relu(m[1](m[0](t)))

        blocks = self.decoder.blocks
        out = cbr(blocks[0][1], cbr(blocks[0][0], feats[0]))
        for i in range(1, len(blocks)):
            up = F.interpolate(out, size=feats[i].shape[-2:], mode="bilinear")          # decoder.py:35 (align_corners=False)
            out = cbr(blocks[i][1], cbr(blocks[i][0], torch.cat((up, feats[i]), 1)))     # :36-37, upsampled first
        conv, up, act = self.segmentation_head[0], self.segmentation_head[1], self.segmentation_head[2]
        y = F.conv2d(out, conv.weight, conv.bias, conv.stride, conv.padding)
        if hasattr(up, "scale_factor"):
            y = F.interpolate(y, scale_factor=up.scale_factor, mode="bilinear", align_corners=True)
        return act(y)

    def freeze_encoder(self):
        for p in self.encoder.parameters():
            p.requires_grad = False


def _to_device_layout(x):
    """Model entry: refuse CPU tensors (no fallback) and switch to channels_last (NHWC)."""
    import torch
    if not x.is_cuda:
        raise RuntimeError("vq_seg_amd models run on a 'cuda' (ROCm) device only; there is no CPU fallback "
                           f"(got a tensor on {x.device}).  (The plain `unet` plumbing model has an explicit torch-operator entry "
                           "point for the reference's CPU wiring check: Unet.forward_plumbing.)")
    return x.contiguous(memory_format=torch.channels_last)
