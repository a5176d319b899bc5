"""DarkNet-19 (YOLOv2's backbone) on the MI355X kernels (reference: holocron/models/classification/darknetv2.py).

Same module tree and ``state_dict`` keys as the reference; [Conv2d, BatchNorm2d, LeakyReLU] runs are fused units,
``nn.MaxPool2d(2)`` is ``hc_maxpool2_*``; with ``passthrough=True`` the body also returns the output of the penultimate stage
(darknetv2.py:139-151; no clone is needed: nothing here writes activations in place).
"""
from collections import OrderedDict
from typing import Any, Callable, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import _lib
from ...nn import GlobalAvgPool2d
from ...nn.convbn_op import prepack_model_convs, run_conv_sequence
from ...nn.init import init_module
from ...nn.repblock_op import POOL
from ..utils import conv_sequence
from .darknet import _FusedSequential

__all__ = ["DarknetBodyV2", "DarknetV2", "darknet19"]


class DarknetBodyV2(nn.Sequential):
    def __init__(self, layout: List[Tuple[int, int]], in_channels: int = 3, stem_channels: int = 32, passthrough: bool = False,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None, conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        if act_layer is None:
            act_layer = nn.LeakyReLU(0.1, inplace=True)
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        in_chans = [stem_channels] + [_layout[0] for _layout in layout[:-1]]
        super().__init__(OrderedDict([
            ("stem", _FusedSequential(*conv_sequence(in_channels, stem_channels, act_layer, norm_layer, drop_layer, conv_layer,
                                                     kernel_size=3, padding=1, bias=(norm_layer is None)))),
            ("layers", nn.Sequential(*[self._make_layer(num_blocks, _in_chans, out_chans, act_layer, norm_layer, drop_layer, conv_layer)
                                       for _in_chans, (out_chans, num_blocks) in zip(in_chans, layout)])),
        ]))
        self.passthrough = passthrough

    @staticmethod
    def _make_layer(num_blocks: int, in_planes: int, out_planes: int, act_layer: Optional[nn.Module] = None,
                    norm_layer: Optional[Callable[[int], nn.Module]] = None, drop_layer: Optional[Callable[..., nn.Module]] = None,
                    conv_layer: Optional[Callable[..., nn.Module]] = None) -> nn.Sequential:
        layers: List[nn.Module] = [nn.MaxPool2d(2)]
        layers.extend(conv_sequence(in_planes, out_planes, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1,
                                    stride=1, bias=(norm_layer is None)))
        for _ in range(num_blocks):
            layers.extend(
                conv_sequence(out_planes, out_planes // 2, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=1, padding=0,
                              stride=1, bias=(norm_layer is None))
                + conv_sequence(out_planes // 2, out_planes, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1,
                                stride=1, bias=(norm_layer is None)))
        return _FusedSequential(*layers)

    def forward(self, x: torch.Tensor) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:  # type: ignore[override]
        if self.passthrough:
            x = self.stem(x)
            aux = x
            for idx, layer in enumerate(self.layers):
                x = layer(x)
                if idx == len(self.layers) - 2:
                    aux = x
            return x, aux
        return super().forward(x)


class DarknetV2(nn.Sequential):
    def __init__(self, layout: List[Tuple[int, int]], num_classes: int = 10, in_channels: int = 3, stem_channels: int = 32,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None, conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__(OrderedDict([
            ("features", DarknetBodyV2(layout, in_channels, stem_channels, False, act_layer, norm_layer, drop_layer, conv_layer)),
            ("classifier", nn.Conv2d(layout[-1][0], num_classes, 1)),
            ("pool", GlobalAvgPool2d(flatten=True)),
        ]))
        init_module(self, "leaky_relu")

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        _lib.require_gpu(x)
        prepack_model_convs(self)
        POOL.begin(x.device)
        try:
            h = self.features(x)
            h = run_conv_sequence([self.classifier], h, padded_out=True)       # 1x1 conv + bias, channel-padded logits
            pooled = self.pool(h)
            return pooled[:, :self.classifier.out_channels]
        finally:
            POOL.end()


def darknet19(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> DarknetV2:
    """Darknet-19 (darknetv2.py:214-237)."""
    if pretrained or checkpoint is not None:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")
    return DarknetV2([(64, 0), (128, 1), (256, 1), (512, 2), (1024, 2)], **kwargs)
