"""DarkNet-24 (YOLOv1's backbone) on the MI355X kernels (reference: holocron/models/classification/darknet.py).

Same module tree and ``state_dict`` keys as the reference (``features.stem.*``, ``features.layers.<stage>.<i>.*``,
``classifier.*``).  Every [Conv2d, BatchNorm2d?, LeakyReLU] run is one fused unit (with ``norm_layer=None`` the bias and the
activation ride in the gather-conv epilogue), ``nn.MaxPool2d(2)`` is ``hc_maxpool2_*``, the 7x7 3-channel stem goes through the
im2col column tensor.
"""
from collections import OrderedDict
from typing import Any, Callable, List, Optional

import torch
import torch.nn as nn

from ... import _lib
from ...nn import GlobalAvgPool2d
from ...nn.convbn_op import prepack_model_convs, run_conv_sequence
from ...nn.init import init_module
from ...nn.repblock_op import POOL
from ..utils import conv_sequence

__all__ = ["DarknetBodyV1", "DarknetV1", "darknet24"]


class _FusedSequential(nn.Sequential):
    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        return run_conv_sequence(self, x)


class DarknetBodyV1(nn.Sequential):
    def __init__(self, layout: List[List[int]], in_channels: int = 3, stem_channels: int = 64, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None, drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        if act_layer is None:
            act_layer = nn.LeakyReLU(0.1, inplace=True)
        in_chans = [stem_channels] + [_layout[-1] for _layout in layout[:-1]]
        super().__init__(OrderedDict([
            ("stem", _FusedSequential(*conv_sequence(in_channels, stem_channels, act_layer, norm_layer, drop_layer, conv_layer,
                                                     kernel_size=7, padding=3, stride=2, bias=(norm_layer is None)))),
            ("layers", nn.Sequential(*[self._make_layer([_in_chans, *planes], act_layer, norm_layer, drop_layer, conv_layer)
                                       for _in_chans, planes in zip(in_chans, layout)])),
        ]))
        init_module(self, "leaky_relu")

    @staticmethod
    def _make_layer(planes: List[int], act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                    drop_layer: Optional[Callable[..., nn.Module]] = None,
                    conv_layer: Optional[Callable[..., nn.Module]] = None) -> nn.Sequential:
        layers: List[nn.Module] = [nn.MaxPool2d(2)]
        for in_planes, out_planes in zip(planes[:-1], planes[1:]):
            layers.extend(conv_sequence(in_planes, out_planes, act_layer, norm_layer, drop_layer, conv_layer,
                                        kernel_size=3 if out_planes > in_planes else 1, padding=1 if out_planes > in_planes else 0,
                                        bias=(norm_layer is None)))
        return _FusedSequential(*layers)


class DarknetV1(nn.Sequential):
    def __init__(self, layout: List[List[int]], num_classes: int = 10, in_channels: int = 3, stem_channels: int = 64,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None, conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__(OrderedDict([
            ("features", DarknetBodyV1(layout, in_channels, stem_channels, act_layer, norm_layer, drop_layer, conv_layer)),
            ("pool", GlobalAvgPool2d(flatten=True)),
            ("classifier", nn.Linear(layout[2][-1], num_classes)),
        ]))
        init_module(self, "leaky_relu")

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        _lib.require_gpu(x)
        prepack_model_convs(self)
        POOL.begin(x.device)
        try:
            return super().forward(x)
        finally:
            POOL.end()


def darknet24(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> DarknetV1:
    """Darknet-24 (darknet.py:137-160)."""
    if pretrained or checkpoint is not None:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")
    return DarknetV1([[192], [128, 256, 256, 512], [*([256, 512] * 4), 512, 1024], [512, 1024] * 2], **kwargs)
