"""Parameter initialisation with the reference's draws (holocron/nn/init.py:10-24): walking ``module.modules()`` in order, every
convolution takes one Kaiming-normal (fan-out) draw from the global generator, normalisation layers are reset to the identity
affine and ``nn.Linear`` keeps torch's constructor initialisation - so a seeded build reproduces the reference's parameters."""
import torch
from torch import nn
from torch.nn.modules.conv import _ConvNd

__all__ = ["init_module"]

_NORM_LAYERS = (nn.BatchNorm2d, nn.GroupNorm)


@torch.no_grad()
def _reset_conv(layer: _ConvNd, nonlinearity: str) -> None:
    nn.init.kaiming_normal_(layer.weight, mode="fan_out", nonlinearity=nonlinearity)
    if layer.bias is not None:
        layer.bias.zero_()


@torch.no_grad()
def _reset_norm(layer: nn.Module) -> None:
    layer.weight.fill_(1.0)
    layer.bias.zero_()


def init_module(module: nn.Module, nonlinearity: str = "relu") -> None:
    for layer in module.modules():
        if isinstance(layer, _ConvNd):
            _reset_conv(layer, nonlinearity)
        elif isinstance(layer, _NORM_LAYERS):
            _reset_norm(layer)
