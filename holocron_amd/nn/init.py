"""Parameter initialisation (reference: holocron/nn/init.py:10-24)."""
import torch.nn as nn
from torch.nn.modules.conv import _ConvNd

__all__ = ["init_module"]


def init_module(module: nn.Module, nonlinearity: str = "relu") -> None:
    """Kaiming-normal (fan_out) for every conv, unit scale / zero shift for BN and GroupNorm.
    ``nn.Linear`` keeps torch's default initialisation, like the reference."""
    for m in module.modules():
        if isinstance(m, _ConvNd):
            nn.init.kaiming_normal_(m.weight.data, mode="fan_out", nonlinearity=nonlinearity)
            if m.bias is not None:
                m.bias.data.zero_()
            continue
        if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            m.weight.data.fill_(1.0)
            m.bias.data.zero_()
