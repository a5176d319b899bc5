"""CSP-Darknet-53 on the MI355X kernels (reference: holocron/models/classification/darknetv4.py).

Same module tree and ``state_dict`` keys as the reference (``features.stem.*``,
``features.stages.<i>.{base_layer,main,transition}.*``, ``classifier.*``).  Every
[Conv2d, BatchNorm2d, act, DropBlock2d?] run is one fused conv_bn_act call; the CSP split / concat
(darknetv4.py:112-115) is two channel-slice copies, with the main branch writing its result straight into
the concat buffer.
"""
from collections import OrderedDict
import os
from typing import Any, Callable, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import _lib
from ...nn import GlobalAvgPool2d
from ...nn.convbn_op import prepack_model_convs, run_conv_sequence
from ...nn.functional import drop_plan_scope
from ...nn.init import init_module
from ...nn.repblock_op import POOL
from ...ops.nhwc import cat_buffer, cat_cl, chunk2_cl, slice_of, split_keep_cl
from ..utils import conv_sequence
from .darknetv3 import ResBlock

__all__ = ["CSPStage", "DarknetBodyV4", "DarknetV4", "cspdarknet53", "cspdarknet53_mish"]


class CSPStage(nn.Module):
    """Cross-stage-partial stage (darknetv4.py:37-115): stride-2 3x3 + 1x1 "base", half of the channels through
    ``num_blocks`` residual blocks + 1x1, concat with the other half, 1x1 transition."""

    def __init__(self, in_channels: int, out_channels: int, num_blocks: int = 1, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__()
        compression = 2 if num_blocks > 1 else 1
        bias = norm_layer is None
        width = out_channels // compression
        self.base_layer = nn.Sequential(
            *conv_sequence(in_channels, out_channels, act_layer, norm_layer, drop_layer, conv_layer,
                           kernel_size=3, padding=1, stride=2, bias=bias),
            *conv_sequence(out_channels, 2 * width, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=1, bias=bias),
        )
        self.main = nn.Sequential(
            *[ResBlock(width, width if num_blocks > 1 else in_channels, act_layer, norm_layer, drop_layer, conv_layer)
              for _ in range(num_blocks)],
            *conv_sequence(width, width, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=1, bias=bias),
        )
        self.transition = nn.Sequential(
            *conv_sequence(2 * width, out_channels, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=1, bias=bias)
        )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if os.environ.get("HC_CSP_SPLIT", "1") == "0":          # A/B: three copies (x -> a, x -> b, a -> buffer)
            x = run_conv_sequence(self.base_layer, x)
            x1, x2 = chunk2_cl(x)
            N, half, H, W = x2.shape
            buf, (_, p2) = cat_buffer(N, [half, half], H, W, x.device)
            y2 = run_conv_sequence(self.main, x2, out=p2)
            return run_conv_sequence(self.transition, cat_cl([x1, y2], buf))
        # the base layer writes straight into the concat buffer of the transition: its first half is already where the concat wants
        # it, the second is copied out for the main path, whose output then takes its place (ops/nhwc.py: split_keep_cl)
        N, _, H, W = x.shape
        H2, W2 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        half = self.transition[0].in_channels // 2
        buf, (_, p2) = cat_buffer(N, [half, half], H2, W2, x.device)
        x = run_conv_sequence(self.base_layer, x, out=slice_of(buf, 0, 2 * half))
        x1, x2 = split_keep_cl(x)
        y2 = run_conv_sequence(self.main, x2, out=p2)
        return run_conv_sequence(self.transition, cat_cl([x1, y2], buf))

    forward_hip = forward


class DarknetBodyV4(nn.Sequential):
    def __init__(self, layout: List[Tuple[int, int]], in_channels: int = 3, stem_channels: int = 32, num_features: int = 1,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        act_layer = nn.LeakyReLU(inplace=True) if act_layer is None else act_layer
        norm_layer = nn.BatchNorm2d if norm_layer is None else norm_layer
        widths_in = [stem_channels] + [w for w, _ in layout[:-1]]
        stem = nn.Sequential(*conv_sequence(in_channels, stem_channels, act_layer, norm_layer, drop_layer, conv_layer,
                                            kernel_size=3, padding=1, bias=(norm_layer is None)))
        stages = nn.Sequential(*[CSPStage(cin, cout, nb, act_layer, norm_layer, drop_layer, conv_layer)
                                 for cin, (cout, nb) in zip(widths_in, layout)])
        super().__init__(OrderedDict([("stem", stem), ("stages", stages)]))
        self.num_features = num_features

    def forward(self, x: torch.Tensor) -> Union[torch.Tensor, List[torch.Tensor]]:  # type: ignore[override]
        _lib.require_gpu(x)
        x = run_conv_sequence(self.stem, x)
        feats = []
        for idx, stage in enumerate(self.stages):
            x = stage(x)
            if idx >= len(self.stages) - self.num_features:
                feats.append(x)
        return x if self.num_features == 1 else feats


class DarknetV4(nn.Sequential):
    def __init__(self, layout: List[Tuple[int, int]], num_classes: int = 10, in_channels: int = 3, stem_channels: int = 32,
                 num_features: int = 1, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__(OrderedDict([
            ("features", DarknetBodyV4(layout, in_channels, stem_channels, num_features, act_layer, norm_layer, drop_layer,
                                       conv_layer)),
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
            with drop_plan_scope(self, x.device):
                return super().forward(x)
        finally:
            POOL.end()


_LAYOUT = [(64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)]


def _no_download(pretrained, checkpoint):
    if pretrained or checkpoint is not None:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")


def cspdarknet53(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> DarknetV4:
    """CSP-Darknet-53 (darknetv4.py:256-281), LeakyReLU(0.01) activations."""
    _no_download(pretrained, checkpoint)
    return DarknetV4(_LAYOUT, **kwargs)


def cspdarknet53_mish(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> DarknetV4:
    """CSP-Darknet-53 with Mish activations and DropBlock (darknetv4.py:284-319)."""
    _no_download(pretrained, checkpoint)
    from ...nn import DropBlock2d
    kwargs["act_layer"] = nn.Mish(inplace=True)
    kwargs["drop_layer"] = DropBlock2d
    return DarknetV4(_LAYOUT, **kwargs)
