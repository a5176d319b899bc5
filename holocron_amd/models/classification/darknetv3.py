"""DarkNet-53 on the MI355X kernels (reference: holocron/models/classification/darknetv3.py,
holocron/models/classification/resnet.py:59-87).

Same module tree and ``state_dict`` keys as the reference (``features.stem.*``,
``features.layers.<stage>.<i>.*``, ``features.layers.<stage>.<block>.conv.{0,1,3,4}.*``,
``classifier.*``); every [Conv2d, BatchNorm2d, LeakyReLU] run executes as one fused
conv_bn_act call (holocron_amd/nn/convbn_op.py), the residual add rides in its apply pass.
"""
from collections import OrderedDict
from typing import Any, Callable, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import _lib
from ...nn import DropBlock2d, GlobalAvgPool2d
from ...nn.convbn_op import prepack_model_convs, run_conv_sequence
from ...nn.init import init_module
from ...nn.repblock_op import POOL
from ..utils import conv_sequence

__all__ = ["ResBlock", "DarknetBodyV3", "DarknetV3", "darknet53"]


class _FusedSequential(nn.Sequential):
    """nn.Sequential whose conv/bn/act runs execute fused on the HIP path."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        return run_conv_sequence(self, x)


class ResBlock(nn.Module):
    """1x1 -> 3x3 bottleneck with identity shortcut and NO activation after the add
    (darknetv3.py:23-70 on top of resnet.py:59-87 with downsample=None, act_layer=None)."""

    def __init__(self, planes: int, mid_planes: int, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__()
        bias = norm_layer is None
        self.conv = nn.Sequential(
            *conv_sequence(planes, mid_planes, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=1, bias=bias),
            *conv_sequence(mid_planes, planes, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1, bias=bias),
        )
        self.downsample = None
        if drop_layer is not None:
            self.dropblock = DropBlock2d(0.1, 7, inplace=True)
        if hasattr(self.conv[-1], "inplace"):
            self.conv[-1].inplace = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # dropblock(x + conv(x)) (darknetv3.py:59-61): the DropBlock behind the residual add rides in the last unit's BatchNorm passes
        # (hc_bn_act_*_post) instead of being a read + write of the block output of its own, forward and backward
        return run_conv_sequence(self.conv, x, residual=x, post_drop=getattr(self, "dropblock", None))

    forward_hip = forward


class DarknetBodyV3(nn.Sequential):
    def __init__(self, layout: List[Tuple[int, int]], in_channels: int = 3, stem_channels: int = 32,
                 num_features: int = 1, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        act_layer = nn.LeakyReLU(0.1, inplace=True) if act_layer is None else act_layer
        norm_layer = nn.BatchNorm2d if norm_layer is None else norm_layer
        widths_in = [stem_channels] + [w for w, _ in layout[:-1]]
        bias = norm_layer is None
        stem = _FusedSequential(*conv_sequence(in_channels, stem_channels, act_layer, norm_layer, drop_layer, conv_layer,
                                               kernel_size=3, padding=1, bias=bias))
        stages = [self._make_layer(nb, cin, cout, act_layer, norm_layer, drop_layer, conv_layer)
                  for cin, (cout, nb) in zip(widths_in, layout)]
        super().__init__(OrderedDict([("stem", stem), ("layers", nn.Sequential(*stages))]))
        self.num_features = num_features

    @staticmethod
    def _make_layer(num_blocks: int, in_planes: int, out_planes: int, act_layer=None, norm_layer=None, drop_layer=None,
                    conv_layer=None) -> nn.Sequential:
        """stride-2 3x3 conv, then ``num_blocks`` residual blocks (darknetv3.py:126-151)."""
        layers: List[nn.Module] = conv_sequence(in_planes, out_planes, act_layer, norm_layer, drop_layer, conv_layer,
                                                kernel_size=3, padding=1, stride=2, bias=(norm_layer is None))
        layers += [ResBlock(out_planes, out_planes // 2, act_layer, norm_layer, drop_layer, conv_layer)
                   for _ in range(num_blocks)]
        return _FusedSequential(*layers)

    def forward(self, x: torch.Tensor) -> Union[torch.Tensor, List[torch.Tensor]]:  # type: ignore[override]
        if self.num_features == 1:
            return super().forward(x)
        x = self.stem(x)
        feats = []
        for idx, stage in enumerate(self.layers):
            x = stage(x)
            if idx >= len(self.layers) - self.num_features:
                feats.append(x)
        return feats


class DarknetV3(nn.Sequential):
    def __init__(self, layout: List[Tuple[int, int]], num_classes: int = 10, in_channels: int = 3, stem_channels: int = 32,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__(OrderedDict([
            ("features", DarknetBodyV3(layout, in_channels, stem_channels, 1, act_layer, norm_layer, drop_layer, conv_layer)),
            ("pool", GlobalAvgPool2d(flatten=True)),
            ("classifier", nn.Linear(layout[-1][0], num_classes)),
        ]))
        init_module(self, "leaky_relu")
        self.default_cfg = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        _lib.require_gpu(x)
        prepack_model_convs(self)
        POOL.begin(x.device)
        try:
            return super().forward(x)
        finally:
            POOL.end()


def darknet53(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> DarknetV3:
    """Darknet-53 (darknetv3.py:221-250): layout [(64,1),(128,2),(256,8),(512,8),(1024,4)]."""
    if pretrained or checkpoint is not None:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")
    return DarknetV3([(64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)], **kwargs)
