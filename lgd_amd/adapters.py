"""Adapters applied to the student features before the distillation loss
[ref: models/adapters/build.py:10-17, models/adapters/sequential_convs.py:8-15]."""
import torch
from torch import nn

from . import ops
from .registry import ADAPTERS_REGISTRY


@ADAPTERS_REGISTRY.register()
class SequentialConvs(nn.Module):
    """conv3x3 - ReLU - conv3x3 - ReLU - conv3x3, 256 channels (state_dict: adapter.{0,2,4}.*)."""

    def __init__(self, cfg) -> None:
        super().__init__()
        layers = []
        for i in range(3):
            layers.append(ops.Conv3x3(256, 256))
            if i < 2:
                layers.append(nn.ReLU())
        self.adapter = nn.Sequential(*layers)

    def forward(self, x):
        return self.adapter(x)

    def levels(self, xs):
        """the adapter on a list of pyramid levels: each conv is one pass over the concatenated levels, ReLU fused."""
        a = self.adapter
        return ops.conv3x3_chain(xs, [(a[i].weight, a[i].bias) for i in (0, 2, 4)], (True, True, False))


def build_adapter(cfg):
    model = ADAPTERS_REGISTRY.get(cfg.MODEL.DISTILLATOR.ADAPTER.META_ARCH)(cfg)
    return model.to(torch.device(cfg.MODEL.DEVICE))
