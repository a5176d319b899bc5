"""Model-building helpers (reference: holocron/models/utils.py:28-86, 116-143)."""
import logging
from typing import Any, Callable, List, Optional, Tuple

import torch
from torch import nn

__all__ = ["conv_sequence", "fuse_conv_bn"]

logger = logging.getLogger(__name__)


def conv_sequence(
    in_channels: int,
    out_channels: int,
    act_layer: Optional[nn.Module] = None,
    norm_layer: Optional[Callable[[int], nn.Module]] = None,
    drop_layer: Optional[Callable[..., nn.Module]] = None,
    conv_layer: Optional[Callable[..., nn.Module]] = None,
    bn_channels: Optional[int] = None,
    attention_layer: Optional[Callable[[int], nn.Module]] = None,
    blurpool: bool = False,
    **kwargs: Any,
) -> List[nn.Module]:
    """[conv, norm?, act?, blurpool?, attention?, drop?] in the reference's order
    (holocron/models/utils.py:61-84).  The conv has a bias only when no norm follows."""
    make_conv = nn.Conv2d if conv_layer is None else conv_layer
    width = out_channels if bn_channels is None else bn_channels
    stride = kwargs.get("stride", 1)
    downsample_with_blur = blurpool and stride > 1
    if downsample_with_blur:
        kwargs["stride"] = 1
    kwargs.setdefault("bias", norm_layer is None)
    seq: List[nn.Module] = [make_conv(in_channels, out_channels, **kwargs)]
    if callable(norm_layer):
        seq.append(norm_layer(width))
    if callable(act_layer):
        seq.append(act_layer)
    if downsample_with_blur:
        raise NotImplementedError("BlurPool2d is outside the MI355X hot path (SURVEY.md §2)")
    if callable(attention_layer):
        seq.append(attention_layer(width))
    if callable(drop_layer):
        seq.append(drop_layer(inplace=True))
    return seq


def fuse_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fold an eval-mode BatchNorm into the preceding convolution: returns (kernel, bias) with
    kernel = W * gamma / sqrt(running_var + eps), bias = beta - running_mean * gamma / sqrt(...)
    (reference: holocron/models/utils.py:116-143)."""
    if bn.bias.data.shape[0] != conv.weight.data.shape[0]:
        raise AssertionError("expected same number of output channels for both `conv` and `bn`")
    scale = bn.weight.data / torch.sqrt(bn.running_var + bn.eps)
    bias = bn.bias.data - scale * bn.running_mean
    if conv.bias is not None:
        logger.warning("convolution layers placed before batch normalization should not have a bias.")
        bias += scale * conv.bias.data
    kernel = scale.view(-1, 1, 1, 1) * conv.weight.data
    return kernel, bias
