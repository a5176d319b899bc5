"""``Resize`` (squish / aspect-preserving pad) and ``RandomZoomOut`` of the reference's input pipeline (holocron/transforms/
interpolation.py) for CHW **tensors** - on the device when the tensor is there - through ``F.interpolate`` / ``F.pad``.  The
reference builds them on torchvision (PIL or tensor input); torchvision is not part of this image, so PIL input raises."""
from enum import Enum
from math import sqrt
from typing import Any, Tuple

import torch
import torch.nn.functional as F
from torch import nn

__all__ = ["RandomZoomOut", "Resize", "ResizeMethod"]


class ResizeMethod(str, Enum):
    SQUISH = "squish"
    PAD = "pad"


def _shape(image) -> Tuple[int, int]:
    if not isinstance(image, torch.Tensor):
        raise TypeError("expected arg 'image' to be a torch.Tensor (PIL input needs torchvision, which this image does not ship)")
    if image.ndim != 3:
        raise ValueError("the input tensor is expected to be 3-dimensional")
    return image.shape[1], image.shape[2]


def _check_size(size) -> None:
    if not isinstance(size, (tuple, list)) or len(size) != 2 or any(s <= 0 for s in size):
        raise ValueError("size is expected to be a sequence of 2 positive integers")


def _resize(image: torch.Tensor, hw: Tuple[int, int], mode: str) -> torch.Tensor:
    kw = {"align_corners": False, "antialias": True} if mode in ("bilinear", "bicubic") else {}
    out = F.interpolate(image[None].float(), size=hw, mode=mode, **kw)[0]
    return out.to(image.dtype) if image.dtype.is_floating_point else out.round().to(image.dtype)


def _pad_to(image: torch.Tensor, size: Tuple[int, int], pad_mode: str, offset=None) -> torch.Tensor:
    h, w = image.shape[1:]
    hp, wp = size[0] - h, size[1] - w
    top, left = (hp // 2, wp // 2) if offset is None else offset
    pads = (left, wp - left, top, hp - top)
    if pad_mode == "constant":
        return F.pad(image, pads)
    return F.pad(image[None].float(), pads, mode=pad_mode)[0].to(image.dtype)


class Resize(nn.Module):
    """interpolation.py:41-101: ``mode="squish"`` resizes to ``size``; ``mode="pad"`` keeps the aspect ratio and pads."""

    def __init__(self, size: Tuple[int, int], mode: ResizeMethod = ResizeMethod.SQUISH, pad_mode: str = "constant",
                 interpolation: str = "bilinear", **kwargs: Any) -> None:
        if not isinstance(mode, ResizeMethod):
            raise ValueError("mode is expected to be a ResizeMethod")
        _check_size(size)
        super().__init__()
        self.size, self.mode, self.pad_mode, self.interpolation = tuple(size), mode, pad_mode, interpolation

    def get_params(self, image: torch.Tensor) -> Tuple[int, int]:
        h, w = _shape(image)
        ratio = h / w
        if self.size[0] / self.size[1] > ratio:
            return round(self.size[1] * ratio), self.size[1]
        return self.size[0], round(self.size[0] / ratio)

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        _shape(image)
        if self.mode == ResizeMethod.SQUISH:
            return _resize(image, self.size, self.interpolation)
        return _pad_to(_resize(image, self.get_params(image), self.interpolation), self.size, self.pad_mode)


class RandomZoomOut(nn.Module):
    """interpolation.py:104-156: shrink the image to a random fraction of the target area and paste it at a random position
    of a ``size`` canvas."""

    def __init__(self, size: Tuple[int, int], scale: Tuple[float, float] = (0.5, 1.0), interpolation: str = "bilinear",
                 **kwargs: Any) -> None:
        _check_size(size)
        if len(scale) != 2 or scale[0] > scale[1]:
            raise ValueError("scale is expected to be a couple of floats, the first one being small than the second")
        super().__init__()
        self.size, self.scale, self.interpolation = tuple(size), scale, interpolation

    def get_params(self, image: torch.Tensor) -> Tuple[int, int]:
        h, w = _shape(image)
        frac = (self.scale[1] - self.scale[0]) * torch.rand(1).item() + self.scale[0]
        aratio, tratio = h / w, self.size[0] / self.size[1]
        max_area = self.size[1] ** 2 * aratio if tratio > aratio else self.size[0] ** 2 / aratio
        area = max_area * frac
        w_ = max(1, round(sqrt(area / aratio)))
        return max(1, min(self.size[0], round(area / w_))), min(self.size[1], w_)

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        h, w = self.get_params(image)
        img = _resize(image, (h, w), self.interpolation)
        top = int(torch.randint(0, self.size[0] - h + 1, (1,)))
        left = int(torch.randint(0, self.size[1] - w + 1, (1,)))
        return _pad_to(img, self.size, "constant", offset=(top, left))
