"""nn.Module wrappers of the hot path (reference: holocron/nn/modules/*.py)."""
from typing import Optional

import torch
from torch import Tensor, nn

from . import functional as F

__all__ = ["HardMish", "GlobalAvgPool2d", "FocalLoss"]


class HardMish(nn.Module):
    """HardMish activation (holocron/nn/modules/activation.py:28-38)."""

    def __init__(self, inplace: bool = False) -> None:
        super().__init__()
        self.inplace = inplace

    def forward(self, x: Tensor) -> Tensor:
        return F.hard_mish(x, inplace=self.inplace)

    def extra_repr(self) -> str:
        return "inplace=True" if self.inplace else ""


class GlobalAvgPool2d(nn.Module):
    """Global average pooling (holocron/nn/modules/downsample.py:57-77)."""

    def __init__(self, flatten: bool = False) -> None:
        super().__init__()
        self.flatten = flatten

    def forward(self, x: Tensor) -> Tensor:
        y = F.global_avg_pool2d(x)
        if self.flatten:
            return y
        return y.view(y.shape[0], y.shape[1], 1, 1)

    def extra_repr(self) -> str:
        return "flatten=True" if self.flatten else ""


class _Loss(nn.Module):
    def __init__(self, weight=None, ignore_index: int = -100, reduction: str = "mean") -> None:
        super().__init__()
        if isinstance(weight, (float, int)):
            self.register_buffer("weight", torch.Tensor([weight, 1 - weight]))
        else:
            self.register_buffer("weight", weight)
        self.ignore_index = ignore_index
        if reduction not in {"none", "mean", "sum"}:
            raise NotImplementedError("argument reduction received an incorrect input")
        self.reduction = reduction


class FocalLoss(_Loss):
    """Focal loss module (holocron/nn/modules/loss.py:50-84)."""

    def __init__(self, gamma: float = 2.0, **kwargs) -> None:
        super().__init__(**kwargs)
        self.gamma = gamma

    def forward(self, x: Tensor, target: Tensor) -> Tensor:
        return F.focal_loss(x, target, self.weight, self.ignore_index, self.reduction, self.gamma)

    def extra_repr(self) -> str:
        return f"gamma={self.gamma}, reduction='{self.reduction}'"
