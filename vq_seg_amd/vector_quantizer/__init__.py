"""make_vq_module / Identity -- same contract as the reference's vector_quantizer/__init__.py:5-32."""
import copy

from torch import nn

from .vq_img import VectorQuantizer, EuclideanCodebook, kmeans, quantize_group  # noqa: F401


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def make_vq_module(vq_cfg, encoder_channels, depth):
    """num_embeddings int -> a VectorQuantizer at every depth; list -> 0 = Identity, >0 = VectorQuantizer with
    dim = encoder_channels[i + 1]; <0 -> ValueError; other type -> TypeError; len != depth -> AssertionError."""
    num = _get(vq_cfg, "num_embeddings")
    rest = {k: v for k, v in dict(vq_cfg).items() if k != "num_embeddings"}
    if isinstance(num, int):
        return nn.ModuleList([VectorQuantizer(num_embeddings=num, **rest, dim=encoder_channels[i + 1])
                              for i in range(depth)])
    if isinstance(num, list):
        assert depth == len(num), "depth and length of vq_cfg.num_embeddings must to be same number"
        layers = []
        for i, n in enumerate(copy.deepcopy(num)):
            if n == 0:
                layers.append(Identity())
            elif n > 0:
                layers.append(VectorQuantizer(num_embeddings=n, **rest, dim=encoder_channels[i + 1]))
            else:
                raise ValueError(f"{n} is not available number of embeddings")
        return nn.ModuleList(layers)
    raise TypeError(f"{type(num)} is not available type")


class Identity(nn.Module):
    def __init__(self):
        super().__init__()
        self.embedding = nn.Identity()

    def forward(self, x):
        return self.embedding(x), None, None, None
